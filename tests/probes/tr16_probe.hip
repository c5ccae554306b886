// Hardware probe (GPU box only): pins down the per-lane-address semantics of ds_read_b64_tr_b16 that
// hydragen_amd/csrc/prefix_attn.hip relies on for the V^T MFMA operand.
//   hipcc --offload-arch=gfx950 -O2 tests/probes/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_ptr;

// mode 0: linear addresses (lane*8 bytes).  mode 1: "row (i16>>2) of a 256-byte-stride matrix, columns
// 4*(i16&3).. of the 16-column block chosen by (g16&1)", rows offset by 4*(lane>>5)
__global__ void probe(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, i16 = lane & 15, g16 = lane >> 4, hi = lane >> 5;
    int byte_off;
    if (mode == 0) byte_off = lane * 8;
    else byte_off = (4 * hi + (i16 >> 2)) * 256 + 32 * (g16 & 1) + 8 * (i16 & 3);
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)((char*)lds + byte_off));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = t[j];
}

int main() {
    short* d;
    hipMalloc(&d, 256 * sizeof(short));
    int bad = 0;
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        std::vector<short> h(256);
        hipMemcpy(h.data(), d, 256 * sizeof(short), hipMemcpyDeviceToHost);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                const int i16 = lane & 15, g16 = lane >> 4, hi = lane >> 5;
                int want;
                if (mode == 0) want = i16 + j * 16 + g16 * 64;
                else want = ((4 * hi + j) * 256 + 32 * (g16 & 1)) / 2 + i16;  // element [row 4hi+j][col 16*(g16&1)+i16]
                if (h[lane * 4 + j] != want) {
                    if (bad < 16) printf("mode %d lane %d elem %d: got %d want %d\n", mode, lane, j, h[lane * 4 + j], want);
                    ++bad;
                }
            }
    }
    printf(bad ? "TR16_PROBE FAIL (%d mismatches)\n" : "TR16_PROBE PASS\n", bad);
    return bad != 0;
}
