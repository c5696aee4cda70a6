// Build shim (ours, not reference code): the reference expects <nlohmann/json.hpp> from its
// 3rd-party/nlohmann_json_cmake_fetchcontent submodule, which is an empty directory in
// /root/reference. The only nlohmann-json on this image is the single-header v3.1.1 below.
#pragma once
#include "/opt/conda/include/json.hpp"
