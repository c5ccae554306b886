// Prefix pass kernel + launcher: the unit body lives in prefix_unit_w64.h, the kernel template in prefix_launch_w64.h.  Replaces flash-attn's _flash_attn_forward / _flash_attn_varlen_forward as called from
// /root/reference/hydragen/flash.py:284-351 and hydragen/attention.py:270,313,344.
// This file: the bf16 instantiations and the dtype dispatch; prefix_attn_w64_f16.hip: the fp16 instantiations.
#include "prefix_launch_w64.h"

namespace hyd {

int launch_prefix_w64(const PrefixArgs& a, int dtype, int D, bool causal, int grid, hipStream_t s) {
#ifdef HYD_ABLATION_BUILD
    if (a.dbg && dtype == HYD_BF16 && D == 128 && !causal && a.wg_rows == 128) {
        switch (a.dbg) {
#define HYD_ABL(N) case N: return launch_prefix_w64_t<BF16, 128, false, 2, N>(a, grid, s);
            HYD_ABL(1) HYD_ABL(4) HYD_ABL(5) HYD_ABL(8) HYD_ABL(32) HYD_ABL(64) HYD_ABL(65) HYD_ABL(128) HYD_ABL(2048) HYD_ABL(4096) HYD_ABL(2049) HYD_ABL(2052) HYD_ABL(2053) HYD_ABL(2056) HYD_ABL(2064) HYD_ABL(8192) HYD_ABL(16384) HYD_ABL(10240) HYD_ABL(18432)
#undef HYD_ABL
            default: break;
        }
    }
#endif
    if (dtype == HYD_F16) return launch_prefix_w64_f16(a, D, causal, grid, s);
    return launch_prefix_w64_dtype<BF16>(a, D, causal, grid, s);
}

}  // namespace hyd
