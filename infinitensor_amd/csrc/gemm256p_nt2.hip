// gemm256p instantiations for NT = 2 column tiles per wave (tile 256 x 128): see gemm256p_kernel.h.
// One translation unit per tile width so that the three compile in parallel.
#include "gemm256p_kernel.h"

namespace irocm {
namespace g256p {

int launch_gemm256p_nt2(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch_p<Bf16Traits, 2>(rt, p, akm, bkm)
                                   : launch_p<F16Traits, 2>(rt, p, akm, bkm);
}

} // namespace g256p
} // namespace irocm
