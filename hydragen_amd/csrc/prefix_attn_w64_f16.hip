// Prefix pass, fp16 instantiations (see prefix_attn_w64.hip / prefix_launch_w64.h).  Replaces flash-attn's _flash_attn_forward /
// _flash_attn_varlen_forward as called from /root/reference/hydragen/flash.py:284-351 and hydragen/attention.py:270,313,344.
#include "prefix_launch_w64.h"

namespace hyd {

int launch_prefix_w64_f16(const PrefixArgs& a, int D, bool causal, int grid, hipStream_t s) {
    return launch_prefix_w64_dtype<F16>(a, D, causal, grid, s);
}

}  // namespace hyd
