// gemm256p instantiations for NT = 4 column tiles per wave (tile 256 x 256): see gemm256p_kernel.h.
// One translation unit per tile width so that the three compile in parallel.
#include "gemm256p_kernel.h"

namespace irocm {
namespace g256p {

int launch_gemm256p_nt4(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch_p<Bf16Traits, 4>(rt, p, akm, bkm)
                                   : launch_p<F16Traits, 4>(rt, p, akm, bkm);
}

// timeline instantiations (bf16, NN layout; NT = 4 here, NT = 3 forwarded) for tools/gemm_timeline.py
int launch_gemm256p_trace(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, int nt, unsigned long long *trace) {
    if (dtype != INFINI_DT_BF16)
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "gemm timeline: bf16 only");
    if (nt == 3)
        return launch_p<Bf16Traits, 3, true>(rt, p, true, false, trace);
    return launch_p<Bf16Traits, 4, true>(rt, p, true, false, trace);
}

} // namespace g256p
} // namespace irocm
