// Perf records of the Device::ROCM kernels that have more than one implementation (MatMul, Conv), and the JSON
// persistence of the PerfEngine — the on-disk autotune cache (reference: MatmulCublasPerfRecordObj
// src/kernels/cuda/matmul.cc:7-23, ConvCuDnnPerfRecordObj src/kernels/cuda/conv.cc:13-34 and
// src/core/perf_engine.cc:7-47). `h.tune()` (RuntimeObj::run(graph, tune = true)) times every candidate kernel of an
// operator on the device and stores the winner under the operator's OpPerfKey; later runs of any graph in the
// process launch that kernel (RocmRuntimeObj::launchAll looks the record up, also for fused launches).
#pragma once
#include "core/kernel.h"

namespace infini {

// record type ids 0, 1, 2 are taken by the reference (perf_engine.cc:5, conv.cc:267, matmul.cc:214)
constexpr int kRocmMatmulRecord = 3;
constexpr int kRocmConvRecord = 4;

struct RocmVariantPerfRecordObj : public PerfRecordObj {
    int recordType = kRocmMatmulRecord;
    int variant = -1; // kernel variant of the C ABI (infini_rocm_matmul_set_variant / infini_rocm_conv2d_set_variant)
    void to_json(json &j) override;
    static PerfRecord from_json(const json &j);
};
using RocmVariantPerfRecord = Ref<RocmVariantPerfRecordObj>;

// the in-memory PerfEngine as JSON text and back (same document layout as perf_engine.cc:23-45)
std::string perfEngineToJson();
void perfEngineFromJson(const std::string &text);

} // namespace infini
