// Build shim: json 3.1.1 has no separate forward header.
#pragma once
#include <nlohmann/json.hpp>
