// Detect box decode for one pyramid level: NHWC head conv output -> the reference's permuted raw
// tensor x[i] (B, na, ny, nx, no) and the decoded rows of z (B, rows, no), both fp16.
//
// A workgroup stages TP consecutive pixels of one image (all na*no head channels, 16-byte loads)
// in LDS.  For a fixed anchor the TP*no outputs of those pixels are contiguous in x[i] and in z,
// so each lane then produces 8 consecutive output elements, looks their sources up in LDS and
// issues one 16-byte store to each tensor (2-byte stores when the run is not 16-byte aligned).
#include "y5_common.h"

namespace {

#include "decode_tile.h"

__global__ __launch_bounds__(256) void sigmoid_kernel(const f16* __restrict__ x, f16* __restrict__ y, long long n, int libm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = libm ? sigmoid_h_libm(x[i]) : sigmoid_h(x[i]);
}

template <bool VEC, typename ET>
__global__ __launch_bounds__(kThreads) void detect_decode_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    decode_tile<VEC, ET>(a, blockIdx.x, reinterpret_cast<f16*>(smem_raw));
}

// All pyramid levels in one launch: the small levels (a few hundred workgroups) run beside the large one instead
// of each paying its own ramp-up and tail.
struct DecodeMultiArgs {
    DecodeArgs lv[Y5_MAX_LEVELS];
    int first_block[Y5_MAX_LEVELS + 1];  // workgroups [first_block[l], first_block[l+1]) belong to level l
    int n;
};

template <bool VEC, typename ET>
__global__ __launch_bounds__(kThreads) void detect_decode_multi_kernel(const DecodeMultiArgs m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int l = 0;
    while (l + 1 < m.n && (int)blockIdx.x >= m.first_block[l + 1]) ++l;
    decode_tile<VEC, ET>(m.lv[l], blockIdx.x - m.first_block[l], reinterpret_cast<f16*>(smem_raw));
}

}  // namespace

int y5_launch_sigmoid(const void* x, void* y, long long n, int libm, hipStream_t s) {
    Y5_REQUIRE(n >= 0 && (n == 0 || (x && y)), "sigmoid: null pointer");
    if (n == 0) return Y5_OK;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const f16*>(x),
                       static_cast<f16*>(y), n, libm);
    Y5_LAUNCH_CHECK("sigmoid_kernel");
    return Y5_OK;
}

namespace {
// The stand-alone kernels read the head values from memory.
int check_head(const y5_decode_desc& d, const DecodeArgs& a) {
    Y5_REQUIRE(d.head, "decode: null pointer");
    Y5_REQUIRE(d.head_ld % 8 == 0 && d.head_ld >= a.lds_ld, "decode: head_ld %d must be a multiple of 8 >= %d", d.head_ld, a.lds_ld);
    Y5_REQUIRE(y5_aligned16(d.head), "decode: head must be 16-byte aligned");
    return Y5_OK;
}
}  // namespace

int y5_launch_decode(const y5_decode_desc& d, hipStream_t s) {
    DecodeArgs a;
    size_t lds_bytes = 0;
    bool vec = false;
    if (int rc = prepare_level(d, a, &lds_bytes, &vec)) return rc;
    if (int rc = check_head(d, a)) return rc;
    const dim3 grid((unsigned)(a.tiles_per_image * d.batch));
    const bool bf = d.dtype == Y5_BF16;
    if (vec && bf)
        hipLaunchKernelGGL((detect_decode_kernel<true, bf16>), grid, dim3(kThreads), lds_bytes, s, a);
    else if (vec)
        hipLaunchKernelGGL((detect_decode_kernel<true, f16>), grid, dim3(kThreads), lds_bytes, s, a);
    else if (bf)
        hipLaunchKernelGGL((detect_decode_kernel<false, bf16>), grid, dim3(kThreads), lds_bytes, s, a);
    else
        hipLaunchKernelGGL((detect_decode_kernel<false, f16>), grid, dim3(kThreads), lds_bytes, s, a);
    Y5_LAUNCH_CHECK("detect_decode_kernel");
    return Y5_OK;
}

int y5_launch_decode_multi(const y5_decode_desc* d, int n, hipStream_t s) {
    Y5_REQUIRE(d && n >= 1 && n <= Y5_MAX_LEVELS, "decode_multi: %d levels out of range 1..%d", n, Y5_MAX_LEVELS);
    DecodeMultiArgs m{};
    m.n = n;
    size_t lds_max = 0;
    bool vec_all = true;
    long long blocks = 0;
    for (int l = 0; l < n; ++l) {
        size_t lds_bytes = 0;
        bool vec = false;
        if (int rc = prepare_level(d[l], m.lv[l], &lds_bytes, &vec)) return rc;
        if (int rc = check_head(d[l], m.lv[l])) return rc;
        vec_all = vec_all && vec;
        if (lds_bytes > lds_max) lds_max = lds_bytes;
        m.first_block[l] = (int)blocks;
        blocks += (long long)m.lv[l].tiles_per_image * d[l].batch;
        Y5_REQUIRE(blocks < 0x7fffffffLL, "decode_multi: too many tiles");
    }
    m.first_block[n] = (int)blocks;
    const bool bf = d[0].dtype == Y5_BF16;
    for (int l = 1; l < n; ++l) Y5_REQUIRE(d[l].dtype == d[0].dtype, "decode_multi: levels of different element types");
    if (vec_all && bf)
        hipLaunchKernelGGL((detect_decode_multi_kernel<true, bf16>), dim3((unsigned)blocks), dim3(kThreads), lds_max, s, m);
    else if (vec_all)
        hipLaunchKernelGGL((detect_decode_multi_kernel<true, f16>), dim3((unsigned)blocks), dim3(kThreads), lds_max, s, m);
    else if (bf)
        hipLaunchKernelGGL((detect_decode_multi_kernel<false, bf16>), dim3((unsigned)blocks), dim3(kThreads), lds_max, s, m);
    else
        hipLaunchKernelGGL((detect_decode_multi_kernel<false, f16>), dim3((unsigned)blocks), dim3(kThreads), lds_max, s, m);
    Y5_LAUNCH_CHECK("detect_decode_multi_kernel");
    return Y5_OK;
}
