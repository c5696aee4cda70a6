// gemm256p instantiations for NT = 4 column tiles per wave (tile 256 x 256): see gemm256p_kernel.h.
// One translation unit per tile width so that the three compile in parallel.
#include "gemm256p_kernel.h"

namespace irocm {
namespace g256p {

int launch_gemm256p_nt4(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm, int early_a) {
    return dtype == INFINI_DT_BF16 ? launch_p<Bf16Traits, 4>(rt, p, akm, bkm, early_a)
                                   : launch_p<F16Traits, 4>(rt, p, akm, bkm, early_a);
}

} // namespace g256p
} // namespace irocm
