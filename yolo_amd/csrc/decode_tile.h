// Detect box decode of TP staged pixels, shared by the stand-alone decode kernels (decode.hip) and the Detect
// convolution that decodes its own output tile (conv_lw.hip).  Included inside an anonymous namespace.
#pragma once

constexpr int TP = 64;  // pixels per workgroup

constexpr int kThreads = 256;  // threads of the stand-alone decode kernels

struct DecodeArgs {
    const f16* head;
    f16* x_out;
    f16* z_out;
    int head_ld, ny, nx, na, no;
    int z_rows_total, z_row0;
    int tiles_per_image;
    int lds_ld;  // halfs per staged pixel (multiple of 8)
    float stride;
    float anchor_wh[2 * Y5_MAX_ANCHORS];
};

// sigmoid in fp32, rounded once to the element type like a half / bfloat16 tensor's .sigmoid().  The hardware exp2 /
// reciprocal form below gives the same fp16 as 1/(1 + expf(-x)) with IEEE division for every one of the 65536 fp16 inputs
// (checked exhaustively by the tests through y5_sigmoid_f16), at 4 instructions instead of ~25.  bfloat16 keeps 8 bits,
// so the IEEE form is used there: the decode is far from being bound by it.
template <typename ET = f16>
__device__ __forceinline__ ET sigmoid_h(ET v) {
    if constexpr (__is_same(ET, f16)) return (f16)__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-(float)v * 1.44269504088896341f));
    else return (ET)(1.f / (1.f + expf(-(float)v)));
}
__device__ __forceinline__ f16 sigmoid_h_libm(f16 v) { return (f16)(1.f / (1.f + expf(-(float)v))); }

// One decoded element. o: output slot within the anchor record, (gx, gy): cell, (aw, ah): anchor.  Every tensor operation
// of the reference's expression rounds to the element type, except the two that involve the fp32 grid / stride.
template <typename ET = f16>
__device__ __forceinline__ ET decode_one(ET raw, int o, float gx, float gy, float aw, float ah, float stride) {
    const ET s = sigmoid_h<ET>(raw);
    if (o >= 4) return s;
    const ET two_s = (ET)((float)s * 2.f);  // exact doubling, kept as an element-type op
    if (o < 2) {
        const ET c = (ET)((float)two_s - 0.5f);             // element-type result of (y*2 - 0.5)
        const float g = (float)c + (o == 0 ? gx : gy);      // promoted to fp32 by the fp32 grid
        return (ET)(g * stride);                            // * fp32 stride, one rounding on store
    }
    const ET sq = (ET)((float)two_s * (float)two_s);        // (y*2)**2 in the element type
    return (ET)((float)sq * (o == 2 ? aw : ah));            // * anchor_grid (an element-type buffer), element-type result
}

// Decodes np <= TP consecutive pixels (from pix0, a multiple of TP) of image b whose head values lie in LDS at
// tile[p * a.lds_ld + channel]; the caller has synchronised the workgroup (NTH threads) after filling the tile.
// boxdec: (TP + 1) * na * 4 halfs of LDS for the decoded box slots.  Row TP of the tile may be read (never used).
template <bool VEC, int NTH, typename ET = f16>
__device__ __forceinline__ void decode_from_lds(const DecodeArgs& a, int b, int pix0, int np, f16* tile_, f16* boxdec_) {
    constexpr int kThreads = NTH;
    typedef typename Elem<ET>::v8 ev8;
    ET* tile = reinterpret_cast<ET*>(tile_);
    ET* boxdec = reinterpret_cast<ET*>(boxdec_);
    ET* const x_out = reinterpret_cast<ET*>(a.x_out);
    ET* const z_out = reinterpret_cast<ET*>(a.z_out);
    const int npix_img = a.ny * a.nx;

    // The four box slots of every (pixel, anchor) are decoded first, densely (one value per thread), into a small LDS
    // array: inside the 8-wide output windows below they would be a divergent branch that nearly every wave takes.
    // boxdec is [TP + 1][na][4]; the extra row is only ever read, by a window's unused tail
    if (VEC) {
        for (int i = threadIdx.x; i < np * a.na * 4; i += kThreads) {
            const int o = i & 3, pa = i >> 2;
            const int p = pa / a.na, an = pa - p * a.na;
            const int pix = pix0 + p;
            const int gy = pix / a.nx, gx = pix - gy * a.nx;
            boxdec[i] = decode_one<ET>(tile[p * a.lds_ld + an * a.no + o], o, (float)gx, (float)gy, a.anchor_wh[2 * an],
                                   a.anchor_wh[2 * an + 1], a.stride);
        }
        __syncthreads();
    }

    const int run = np * a.no;  // contiguous outputs per anchor
    for (int an = 0; an < a.na; ++an) {
        const float aw = a.anchor_wh[2 * an], ah = a.anchor_wh[2 * an + 1];
        const long long xbase = (((long long)b * a.na + an) * npix_img + pix0) * a.no;
        const long long zbase = ((long long)b * a.z_rows_total + a.z_row0 + (long long)an * npix_img + pix0) * a.no;
        const ET* col = tile + an * a.no;
        if (VEC) {
            // A window of 8 consecutive outputs lies in pixel p from slot o on and, past that pixel's last slot, in pixel
            // p + 1 from slot 0 on (prepare_level allows this path from no = 6 on).  Every element gets the sigmoid; the few windows that touch a pixel's four
            // box slots are patched from the pre-decoded array afterwards.
            const int skip = a.lds_ld - a.no;
            for (int e0 = threadIdx.x * 8; e0 < run; e0 += kThreads * 8) {
                const int p = e0 / a.no, o = e0 - p * a.no;
                const int n1 = a.no - o;  // elements e >= n1 belong to pixel p + 1
                const ET* src = col + p * a.lds_ld + o;
                ev8 raw, dec;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const ET r = src[e + (e >= n1 ? skip : 0)];  // past the run's end: LDS behind the tile, never stored
                    raw[e] = r;
                    dec[e] = sigmoid_h<ET>(r);
                }
                if (o < 4 || n1 < 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int oe = e < n1 ? o + e : e - n1;
                        const int pe = e < n1 ? p : p + 1;
                        if (oe < 4) dec[e] = boxdec[(pe * a.na + an) * 4 + oe];
                    }
                }
                if (e0 + 8 <= run) {
                    if (x_out) *reinterpret_cast<ev8*>(x_out + xbase + e0) = raw;
                    *reinterpret_cast<ev8*>(z_out + zbase + e0) = dec;
                } else {
                    for (int e = 0; e0 + e < run; ++e) {
                        if (x_out) x_out[xbase + e0 + e] = raw[e];
                        z_out[zbase + e0 + e] = dec[e];
                    }
                }
            }
        } else {
            for (int e0 = threadIdx.x; e0 < run; e0 += kThreads) {
                const int p = e0 / a.no, o = e0 - p * a.no;
                const ET r = col[p * a.lds_ld + o];
                const int pix = pix0 + p;
                const int gy = pix / a.nx, gx = pix - gy * a.nx;
                if (x_out) x_out[xbase + e0] = r;
                z_out[zbase + e0] = decode_one<ET>(r, o, (float)gx, (float)gy, aw, ah, a.stride);
            }
        }
    }
}

// One workgroup's work: tile `block` (TP pixels of one image) of the level described by a, staged from a.head.
template <bool VEC, typename ET = f16>
__device__ __forceinline__ void decode_tile(const DecodeArgs& a, int block, f16* tile) {
    const int b = block / a.tiles_per_image;
    const int tp = block - b * a.tiles_per_image;
    const int npix_img = a.ny * a.nx;
    const int pix0 = tp * TP;
    const int np = min(TP, npix_img - pix0);
    const int vec_per_pix = a.lds_ld / 8;
    // np pixels x lds_ld halfs (head_ld >= lds_ld >= chans, so the vector loads stay in the row)
    const f16* src = a.head + ((long long)b * npix_img + pix0) * a.head_ld;
    for (int i = threadIdx.x; i < np * vec_per_pix; i += kThreads) {
        const int p = i / vec_per_pix, v = i - p * vec_per_pix;
        *reinterpret_cast<uint4*>(tile + p * a.lds_ld + v * 8) =
            *reinterpret_cast<const uint4*>(src + (long long)p * a.head_ld + v * 8);
    }
    __syncthreads();
    decode_from_lds<VEC, kThreads, ET>(a, b, pix0, np, tile, tile + TP * a.lds_ld);
}

// Validates one level and fills its kernel arguments; *vec says whether the 16-byte store path applies.
inline int prepare_level(const y5_decode_desc& d, DecodeArgs& a, size_t* lds_bytes, bool* vec) {
    Y5_REQUIRE(d.z_out, "decode: null pointer");
    Y5_REQUIRE(d.dtype == Y5_F16 || d.dtype == Y5_BF16, "decode: dtype %d not in {Y5_F16, Y5_BF16}", d.dtype);
    Y5_REQUIRE(d.batch > 0 && d.ny > 0 && d.nx > 0, "decode: empty tensor");
    Y5_REQUIRE(d.na >= 1 && d.na <= Y5_MAX_ANCHORS, "decode: na %d out of range 1..%d", d.na, Y5_MAX_ANCHORS);
    Y5_REQUIRE(d.no >= 5, "decode: no %d < 5", d.no);
    const int chans = d.na * d.no;
    const int lds_ld = (chans + 7) / 8 * 8;
    Y5_REQUIRE(d.z_row0 >= 0 && (long long)d.z_row0 + (long long)d.na * d.ny * d.nx <= d.z_rows_total, "decode: rows %d..+%d exceed z_rows_total %d", d.z_row0, d.na * d.ny * d.nx, d.z_rows_total);
    *lds_bytes = ((size_t)TP * lds_ld + (size_t)(TP + 1) * d.na * 4) * sizeof(f16);  // staged pixels + decoded box slots
    Y5_REQUIRE(*lds_bytes <= 64 * 1024, "decode: na*no = %d too large for the staging tile", chans);

    a.head = static_cast<const f16*>(d.head);
    a.x_out = static_cast<f16*>(d.x_out);
    a.z_out = static_cast<f16*>(d.z_out);
    a.head_ld = d.head_ld; a.ny = d.ny; a.nx = d.nx; a.na = d.na; a.no = d.no;
    a.z_rows_total = d.z_rows_total; a.z_row0 = d.z_row0;
    const int npix = d.ny * d.nx;
    a.tiles_per_image = (npix + TP - 1) / TP;
    a.lds_ld = lds_ld;
    a.stride = d.stride;
    for (int i = 0; i < 2 * Y5_MAX_ANCHORS; ++i) a.anchor_wh[i] = d.anchor_wh[i];
    Y5_REQUIRE((long long)a.tiles_per_image * d.batch < 0x7fffffffLL, "decode: too many tiles");

    // 16-byte stores need every per-anchor run to start on a multiple of 8 elements in both tensors:
    // runs start at ((b*na + an)*npix + pix0)*no and (b*rows + row0 + an*npix + pix0)*no with pix0 % 64 == 0.
    // and the 8-wide windows assume that a window touches at most pixels p and p + 1, which holds from 6 outputs per
    // anchor on (6: window starts are even; 7: a window ends in slot 6 of p + 1 at the latest).
    *vec = d.no >= 6 && (((long long)npix * d.no) % 8 == 0) && (((long long)d.z_rows_total * d.no) % 8 == 0) &&
           (((long long)d.z_row0 * d.no) % 8 == 0) && y5_aligned16(d.z_out) && y5_aligned16(d.x_out);
    return Y5_OK;
}

