// Batched non-maximum suppression for gfx950, three stages per call and no host round trip:
//   1. candidates: confidence filter + multi-label expansion.  Rows are staged through LDS with 16-byte
//      loads, survivors are compacted with a wave scan and one atomic add per workgroup.
//      Every surviving (anchor row, class) pair becomes one 64-bit sort key
//      [descending-score bits | anchor_row*nc + class]; the slot order in the key buffer is
//      irrelevant because the key itself carries the reference's tie order (lower index first).
//   2. sort: bitonic network over each image's keys.  Images of up to 16384 (padded) keys are sorted by one
//      workgroup per 2048-key chunk, with the cross-chunk passes done in registers through global memory, so a
//      batch fills the chip; larger images fall back to one 1024-thread workgroup per image.  The host does not
//      know the counts: every kernel reads its image's counter and exits at once when it has nothing to do.
//   3. select: one workgroup per image walks the sorted candidates 64 at a time.  The kept boxes
//      live in LDS; 16 waves test the chunk against disjoint parts of the kept list and publish
//      64-bit suppression masks, then wave 0 resolves the chunk greedily with ballots.  Selection
//      stops at max_det kept boxes, which is all the reference returns.
// All IoU arithmetic is IEEE fp32 with the operation order of torchvision's CPU kernel
// (this file is compiled with -ffp-contract=off) and compared in double.
#include "y5_common.h"

namespace {

constexpr int SORT_THREADS = 1024;
constexpr int SORT_LDS_KEYS = 16384;  // 128 KiB of keys per LDS block
constexpr int SEL_THREADS = 1024;
constexpr int SEL_WAVES = SEL_THREADS / 64;
constexpr int MAX_DET_CAP = 1024;
constexpr float CLASS_OFFSET = 4096.f;  // max_wh of the reference

typedef unsigned long long u64;

__device__ __forceinline__ unsigned score_desc_bits(float s) {
    // order-preserving map float -> uint, then inverted so that ascending keys = descending scores
    const unsigned u = __float_as_uint(s);
    const unsigned mono = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~mono;
}
__device__ __forceinline__ float score_from_bits(unsigned inv) {
    const unsigned mono = ~inv;
    const unsigned u = (mono & 0x80000000u) ? (mono & 0x7fffffffu) : ~mono;
    return __uint_as_float(u);
}

__device__ __forceinline__ float ldf(const f16* p) { return (float)*p; }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }

// ------------------------------------------------------------------------------------------
// 1. candidates
// ------------------------------------------------------------------------------------------
// One workgroup stages CAND_ROWS consecutive prediction rows in LDS with 16-byte loads (the rows are contiguous
// in memory), then every thread scores a strided share of the (row, class) pairs from LDS, remembers the
// survivors in a bit mask, the workgroup reserves key slots with ONE atomic add, and the survivors are written.
constexpr int CAND_ROWS = 256;
constexpr int CAND_THREADS = 1024;
static_assert(CAND_THREADS == 4 * CAND_ROWS, "four threads score one staged row");

template <typename T>
__global__ __launch_bounds__(CAND_THREADS) void nms_candidates_kernel(const T* __restrict__ pred, int n_rows, int no, float conf_thres,
                                                                      const uint8_t* __restrict__ class_keep, u64* __restrict__ keys,
                                                                      int cap, int* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    __shared__ int wave_total[CAND_THREADS / 64];
    __shared__ int block_base;

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = no - 5;
    const int r0 = blockIdx.x * CAND_ROWS;
    const int nr = min(CAND_ROWS, n_rows - r0);
    const T* src = pred + ((long long)b * n_rows + r0) * no;
    const int total = nr * no;

    // ---- stage the rows (flat copy)
    constexpr int EPV = 16 / (int)sizeof(T);  // elements per 16-byte vector
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const int nvec = total / EPV;
        for (int i = tid; i < nvec; i += CAND_THREADS) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (int i = nvec * EPV + tid; i < total; i += CAND_THREADS) tile[i] = src[i];
    } else {
        for (int i = tid; i < total; i += CAND_THREADS) tile[i] = src[i];
    }
    __syncthreads();

    u64* out = keys + (long long)b * cap;
    // Four threads per row: thread (row, part) scores classes part, part + 4, ... in chunks of 128 classes, so that a
    // pair's row and class need no division and the row's objectness is read once.
    const int row = tid >> 2, part = tid & 3;
    const bool rvalid = row < nr;
    const float obj = rvalid ? ldf(tile + row * no + 4) : 0.f;
    const bool live = rvalid && obj > conf_thres;
    for (int cbase = 0; cbase < nc; cbase += 128) {
        // ---- pass 1: which of this thread's pairs survive
        unsigned mask = 0;
        if (live) {
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {
                const int c = cbase + part + 4 * i;
                if (c < nc) {
                    const float conf = ldf(tile + row * no + 5 + c) * obj;  // conf = cls * obj in fp32
                    if (conf > conf_thres && (!class_keep || class_keep[c])) mask |= 1u << i;
                }
            }
        }
        // ---- workgroup-wide exclusive scan of the survivor counts, one atomic per workgroup
        const int cnt = __builtin_popcount(mask);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            int sum = 0;
            for (int w = 0; w < CAND_THREADS / 64; ++w) {
                const int v = wave_total[w];
                wave_total[w] = sum;
                sum += v;
            }
            block_base = sum ? atomicAdd(&counters[b], sum) : 0;
        }
        __syncthreads();
        // ---- pass 2: write the survivors' keys (slots past the capacity are dropped; the counter keeps counting)
        int pos = block_base + wave_total[wave] + incl - cnt;
        while (mask) {
            const int i = __builtin_ctz(mask);
            mask &= mask - 1;
            const int c = cbase + part + 4 * i;
            const float conf = ldf(tile + row * no + 5 + c) * obj;
            if (pos < cap) out[pos] = ((u64)score_desc_bits(conf) << 32) | (unsigned)((r0 + row) * nc + c);
            ++pos;
        }
        __syncthreads();  // wave_total / block_base are reused by the next chunk
    }
}

// keys for the plain box-set entry point: one candidate per box
__global__ __launch_bounds__(256) void nms_box_keys_kernel(const float* __restrict__ scores, int n, u64* __restrict__ keys,
                                                           int* __restrict__ counter) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = ((u64)score_desc_bits(scores[i]) << 32) | (unsigned)i;
    if (i == 0) counter[0] = n;
}

// ------------------------------------------------------------------------------------------
// 2. sort (ascending 64-bit keys), one workgroup per image
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void cmpxchg(u64& a, u64& b, bool up) {
    if ((a > b) == up) { const u64 t = a; a = b; b = t; }
}

// One workgroup per image; `min_p2` > 0 makes it skip images whose padded size is below it (those were sorted by the
// chunked kernels further down).
__global__ __launch_bounds__(SORT_THREADS) void nms_sort_kernel(u64* __restrict__ keys_all, int cap_pow2, int cap,
                                                                const int* __restrict__ counters, int min_p2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* s = reinterpret_cast<u64*>(smem_raw);
    const int b = blockIdx.x;
    u64* keys = keys_all + (long long)b * cap_pow2;
    int n = counters[b];
    if (n > cap) n = cap;
    if (n <= 1) return;
    int P = 64;
    while (P < n) P <<= 1;
    if (P < min_p2) return;
    const int tid = threadIdx.x;

    if (P <= SORT_LDS_KEYS) {
        for (int i = tid; i < P; i += SORT_THREADS) s[i] = i < n ? keys[i] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += SORT_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpxchg(s[i], s[i | j], (i & k) == 0);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < n; i += SORT_THREADS) keys[i] = s[i];
        return;
    }

    // large sets: pad in global memory, sort LDS-sized blocks, then merge stages whose partner
    // distance spans blocks through global memory
    for (int i = n + tid; i < P; i += SORT_THREADS) keys[i] = ~0ull;
    __syncthreads();
    const int nblk = P / SORT_LDS_KEYS;
    for (int blk = 0; blk < nblk; ++blk) {
        u64* g = keys + (long long)blk * SORT_LDS_KEYS;
        const int gbase = blk * SORT_LDS_KEYS;
        for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) s[i] = g[i];
        __syncthreads();
        for (int k = 2; k <= SORT_LDS_KEYS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (SORT_LDS_KEYS >> 1); t += SORT_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpxchg(s[i], s[i | j], ((gbase + i) & k) == 0);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) g[i] = s[i];
        __syncthreads();
    }
    for (int k = 2 * SORT_LDS_KEYS; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= SORT_LDS_KEYS; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += SORT_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                u64 x = keys[i], y = keys[i | j];
                if ((x > y) == ((i & k) == 0)) { keys[i] = y; keys[i | j] = x; }
            }
            __syncthreads();
        }
        for (int blk = 0; blk < nblk; ++blk) {
            u64* g = keys + (long long)blk * SORT_LDS_KEYS;
            const int gbase = blk * SORT_LDS_KEYS;
            for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) s[i] = g[i];
            __syncthreads();
            for (int j = SORT_LDS_KEYS >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (SORT_LDS_KEYS >> 1); t += SORT_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpxchg(s[i], s[i | j], ((gbase + i) & k) == 0);
                }
                __syncthreads();
            }
            for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) g[i] = s[i];
            __syncthreads();
        }
    }
}

// Chunked bitonic sort for images of up to CHUNK_MAX_P2 padded keys: one workgroup per 2048-key chunk instead of one
// per image, so that a batch of 32 images fills the chip instead of 32 of its 256 CUs.
//   sort_chunks      every chunk sorted in LDS, alternating direction (stages k <= 2048 of the network)
//   merge_global(k)  passes j = k/2 .. 2048 of stage k: partners in different chunks, 2^levels keys per thread in registers
//   merge_chunks(k)  passes j = 1024 .. 1 of stage k inside each chunk, in LDS
// Images beyond CHUNK_MAX_P2 keys are left to nms_sort_kernel (min_p2 = 2 * CHUNK_MAX_P2).
constexpr int CHUNK = 2048, CHUNK_LOG2 = 11, CHUNK_MAX_P2 = 16384;
static_assert(CHUNK == 2 * 1024 && (1 << CHUNK_LOG2) == CHUNK, "the chunk kernels do one compare-exchange per thread of a 1024-thread workgroup");

__device__ __forceinline__ int padded_size(const int* counters, int b, int cap) {
    int n = counters[b];
    if (n > cap) n = cap;
    int P = 64;
    while (P < n) P <<= 1;
    return n <= 1 ? 0 : P;
}

__global__ __launch_bounds__(1024) void nms_sort_chunks_kernel(u64* __restrict__ keys_all, int cap_pow2, int cap,
                                                               const int* __restrict__ counters) {
    __shared__ u64 s[CHUNK];
    const int b = blockIdx.y, gbase = blockIdx.x * CHUNK, tid = threadIdx.x;
    const int P = padded_size(counters, b, cap);
    if (P == 0 || P > CHUNK_MAX_P2 || gbase >= P) return;
    int n = counters[b];
    if (n > cap) n = cap;
    u64* g = keys_all + (long long)b * cap_pow2 + gbase;
    for (int i = tid; i < CHUNK; i += 1024) s[i] = gbase + i < n ? g[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= CHUNK; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
            cmpxchg(s[i], s[i | j], ((gbase + i) & k) == 0);
            __syncthreads();
        }
    }
    const int lim = P < CHUNK ? P : CHUNK;  // a set smaller than a chunk keeps its own padded size
    for (int i = tid; i < lim; i += 1024) g[i] = s[i];
}

template <int LEVELS>
__global__ __launch_bounds__(256) void nms_merge_global_kernel(u64* __restrict__ keys_all, int cap_pow2, int cap,
                                                               const int* __restrict__ counters, int k) {
    constexpr int G = 1 << LEVELS;
    const int b = blockIdx.y;
    const int P = padded_size(counters, b, cap);
    if (P < k || P > CHUNK_MAX_P2) return;
    const int t = blockIdx.x * 256 + threadIdx.x;  // one thread per group of G keys
    if (t >= P / G) return;
    const int lo = t & (CHUNK - 1), hi = t >> CHUNK_LOG2;
    const int base = (hi << (CHUNK_LOG2 + LEVELS)) | lo;
    u64* g = keys_all + (long long)b * cap_pow2;
    u64 v[G];
#pragma unroll
    for (int m = 0; m < G; ++m) v[m] = g[base | (m << CHUNK_LOG2)];
    const bool up = (base & k) == 0;
#pragma unroll
    for (int bit = LEVELS - 1; bit >= 0; --bit)
#pragma unroll
        for (int m = 0; m < G; ++m)
            if (!(m & (1 << bit))) cmpxchg(v[m], v[m | (1 << bit)], up);
#pragma unroll
    for (int m = 0; m < G; ++m) g[base | (m << CHUNK_LOG2)] = v[m];
}

__global__ __launch_bounds__(1024) void nms_merge_chunks_kernel(u64* __restrict__ keys_all, int cap_pow2, int cap,
                                                                const int* __restrict__ counters, int k) {
    __shared__ u64 s[CHUNK];
    const int b = blockIdx.y, gbase = blockIdx.x * CHUNK, tid = threadIdx.x;
    const int P = padded_size(counters, b, cap);
    if (P < k || P > CHUNK_MAX_P2 || gbase >= P) return;
    u64* g = keys_all + (long long)b * cap_pow2 + gbase;
    s[tid] = g[tid];
    s[tid + 1024] = g[tid + 1024];
    __syncthreads();
    const bool up = (gbase & k) == 0;
    for (int j = CHUNK >> 1; j > 0; j >>= 1) {
        const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
        cmpxchg(s[i], s[i | j], up);
        __syncthreads();
    }
    g[tid] = s[tid];
    g[tid + 1024] = s[tid + 1024];
}

// ------------------------------------------------------------------------------------------
// 3. greedy selection
// ------------------------------------------------------------------------------------------
struct Box { float x1, y1, x2, y2; };

// suppress(j by kept i)  <=>  (double)(inter / (area_i + area_j - inter)) > thr, torchvision CPU order
__device__ __forceinline__ bool suppresses(const Box& k, float karea, const Box& c, float carea, double thr) {
    const float xx1 = fmaxf(k.x1, c.x1), yy1 = fmaxf(k.y1, c.y1);
    const float xx2 = fminf(k.x2, c.x2), yy2 = fminf(k.y2, c.y2);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(karea, carea), inter));
    return (double)ovr > thr;
}

struct SelectArgs {
    const void* pred;      // prediction rows (pred mode) or boxes (n,4) fp32 (box mode)
    const u64* keys;       // sorted keys, cap_pow2 per image
    const int* counters;
    float* det;            // pred mode outputs
    int* count;
    int* kept_index;
    int* n_candidates;
    long long* keep64;     // box mode output
    double iou_thres;
    int n_rows, no, cap, cap_pow2, max_det, agnostic;
};

// One candidate as every lane of every wave holds it: IoU box (class-offset in pred mode), reported box, score, class.
struct Cand {
    Box cb, ob;
    float score, cls, carea;
    unsigned flat;
    bool valid;
};

template <typename T, bool BOXES>
__device__ __forceinline__ Cand load_candidate(const SelectArgs& a, const u64* keys, int b, int ci, int n, int nc) {
    Cand c;
    c.cb = {0.f, 0.f, 0.f, 0.f};
    c.ob = c.cb;
    c.score = c.cls = 0.f;
    c.flat = 0;
    c.valid = ci < n;
    if (c.valid) {
        const u64 key = keys[ci];
        c.flat = (unsigned)key;
        c.score = score_from_bits((unsigned)(key >> 32));
        if (BOXES) {
            const float4 v = reinterpret_cast<const float4*>(a.pred)[c.flat];
            c.cb = {v.x, v.y, v.z, v.w};
            c.ob = c.cb;
        } else {
            const unsigned row = c.flat / (unsigned)nc;
            const unsigned k = c.flat - row * (unsigned)nc;
            const T* p = static_cast<const T*>(a.pred) + ((long long)b * a.n_rows + row) * a.no;
            const float cx = ldf(p), cy = ldf(p + 1), bw = ldf(p + 2), bh = ldf(p + 3);
            const float hw = bw / 2.f, hh = bh / 2.f;
            c.ob = {__fsub_rn(cx, hw), __fsub_rn(cy, hh), __fadd_rn(cx, hw), __fadd_rn(cy, hh)};
            c.cls = (float)k;
            const float off = a.agnostic ? 0.f : __fmul_rn(c.cls, CLASS_OFFSET);
            c.cb = {__fadd_rn(c.ob.x1, off), __fadd_rn(c.ob.y1, off), __fadd_rn(c.ob.x2, off), __fadd_rn(c.ob.y2, off)};
        }
    }
    c.carea = __fmul_rn(__fsub_rn(c.cb.x2, c.cb.x1), __fsub_rn(c.cb.y2, c.cb.y1));
    return c;
}

template <typename T, bool BOXES>
__global__ __launch_bounds__(SEL_THREADS) void nms_select_kernel(const SelectArgs a) {
    static_assert(SEL_WAVES * 4 == 64, "each wave fills four rows of the 64 x 64 chunk matrix");
    __shared__ Box kbox[MAX_DET_CAP];
    __shared__ float karea[MAX_DET_CAP];
    __shared__ u64 sup[SEL_WAVES];
    __shared__ u64 mrow[64];  // mrow[i]: candidates of the chunk that candidate i suppresses if it is kept
    __shared__ int kcount;

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nc = a.no - 5;
    const int total = a.counters[b];
    const int n = total < a.cap ? total : a.cap;
    const u64* keys = a.keys + (long long)b * a.cap_pow2;
    if (threadIdx.x == 0) {
        kcount = 0;
        if (a.n_candidates) a.n_candidates[b] = total;
    }
    __syncthreads();

    // every wave materialises the same 64 candidates; the next chunk is fetched while this one is resolved
    Cand nxt = load_candidate<T, BOXES>(a, keys, b, lane, n, nc);
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int K = kcount;  // stable between the barriers below
        if (K >= a.max_det) break;
        const Cand c = nxt;
        if (c0 + 64 < n) nxt = load_candidate<T, BOXES>(a, keys, b, c0 + 64 + lane, n, nc);

        // (1) kept-list test, kept entries strided over the waves
        bool dead = false;
        for (int k = wave; k < K; k += SEL_WAVES) {
            const Box kb = kbox[k];
            if (!dead && suppresses(kb, karea[k], c.cb, c.carea, a.iou_thres)) dead = true;
        }
        const u64 m = __ballot(dead);
        if (lane == 0) sup[wave] = m;
        // (2) the chunk's own suppression matrix, four rows per wave
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 4 + r;
            Box ib;
            ib.x1 = __shfl(c.cb.x1, i); ib.y1 = __shfl(c.cb.y1, i); ib.x2 = __shfl(c.cb.x2, i); ib.y2 = __shfl(c.cb.y2, i);
            const float iarea = __shfl(c.carea, i);
            const bool sij = lane > i && c.valid && suppresses(ib, iarea, c.cb, c.carea, a.iou_thres);
            const u64 row = __ballot(sij);
            if (lane == 0) mrow[i] = row;
        }
        __syncthreads();

        // (3) wave 0 resolves the chunk greedily on the bit masks and writes the kept candidates
        if (wave == 0) {
            u64 all = 0;
#pragma unroll
            for (int w = 0; w < SEL_WAVES; ++w) all |= sup[w];
            const u64 my = mrow[lane];
            const unsigned my_lo = (unsigned)my, my_hi = (unsigned)(my >> 32);
            u64 alive = __ballot(c.valid) & ~all;
            u64 keep = 0;
            int kk = K;
            while (alive && kk < a.max_det) {
                const int i = __builtin_ctzll(alive);
                keep |= 1ull << i;
                const u64 mi = ((u64)(unsigned)__builtin_amdgcn_readlane((int)my_hi, i) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)my_lo, i);
                alive &= ~(mi | (1ull << i));
                ++kk;
            }
            if ((keep >> lane) & 1) {
                const int pos = K + __builtin_popcountll(keep & ((1ull << lane) - 1));
                kbox[pos] = c.cb;
                karea[pos] = c.carea;
                if (BOXES) {
                    a.keep64[pos] = (long long)c.flat;
                } else {
                    float* d = a.det + ((long long)b * a.max_det + pos) * 6;
                    d[0] = c.ob.x1; d[1] = c.ob.y1; d[2] = c.ob.x2; d[3] = c.ob.y2; d[4] = c.score; d[5] = c.cls;
                    if (a.kept_index) a.kept_index[(long long)b * a.max_det + pos] = (int)c.flat;
                }
            }
            if (lane == 0) kcount = kk;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.count[b] = kcount;
}

// ------------------------------------------------------------------------------------------
// 3b. merge step (utils/utils.py:630-639), one workgroup per image, after the selection: for an image with
// 1 < n < 3000 candidates every kept box becomes the score-weighted mean of ALL candidates whose class-offset box overlaps
// it by more than the threshold (box_iou, utils/utils.py:355-377, fp32, compared in fp32), and kept boxes that no second
// candidate overlaps are dropped ("redundant").  The reference forms the sums with a matrix product whose summation
// order is unspecified; here every sum is accumulated in fp64 in candidate order and rounded once to fp32, then divided
// in fp32 as the reference divides.
// ------------------------------------------------------------------------------------------
constexpr int MERGE_MAX_N = 3000;     // the reference merges only when n < 3E3
constexpr int MERGE_THREADS = 512;
constexpr int MERGE_WAVES = MERGE_THREADS / 64;
struct MergeCand { Box ob; float off, score; };
constexpr size_t MERGE_LDS_BYTES = (size_t)MERGE_MAX_N * sizeof(MergeCand) + (size_t)MAX_DET_CAP * (7 * sizeof(float) + sizeof(int));

template <typename T>
__global__ __launch_bounds__(MERGE_THREADS) void nms_merge_boxes_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
    MergeCand* cand = reinterpret_cast<MergeCand*>(merge_smem);
    float* rows = reinterpret_cast<float*>(cand + MERGE_MAX_N);   // [max_det][6] merged rows
    int* rflat = reinterpret_cast<int*>(rows + MAX_DET_CAP * 6);   // their kept_index entries
    int* alive = rflat + MAX_DET_CAP;                              // 1 = overlapped by a second candidate
    __shared__ int newcount;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = a.no - 5;
    const int n = a.counters[b];
    if (!(n > 1 && n < MERGE_MAX_N) || n > a.cap) return;
    const int K = a.count[b];
    const u64* keys = a.keys + (long long)b * a.cap_pow2;
    for (int i = tid; i < n; i += MERGE_THREADS) {
        const Cand c = load_candidate<T, false>(a, keys, b, i, n, nc);
        cand[i].ob = c.ob;
        cand[i].off = a.agnostic ? 0.f : __fmul_rn(c.cls, CLASS_OFFSET);
        cand[i].score = c.score;
    }
    __syncthreads();
    const float thr = (float)a.iou_thres;
    float* det = a.det + (long long)b * a.max_det * 6;
    for (int k = wave; k < K; k += MERGE_WAVES) {
        const float* r = det + k * 6;
        const float koff = a.agnostic ? 0.f : __fmul_rn(r[5], CLASS_OFFSET);
        const Box kb = {__fadd_rn(r[0], koff), __fadd_rn(r[1], koff), __fadd_rn(r[2], koff), __fadd_rn(r[3], koff)};
        const float karea = __fmul_rn(__fsub_rn(kb.x2, kb.x1), __fsub_rn(kb.y2, kb.y1));
        double sw = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int hits = 0;
        for (int j = lane; j < n; j += 64) {
            const MergeCand c = cand[j];
            const Box cb = {__fadd_rn(c.ob.x1, c.off), __fadd_rn(c.ob.y1, c.off), __fadd_rn(c.ob.x2, c.off), __fadd_rn(c.ob.y2, c.off)};
            const float carea = __fmul_rn(__fsub_rn(cb.x2, cb.x1), __fsub_rn(cb.y2, cb.y1));
            const float w = fmaxf(__fsub_rn(fminf(kb.x2, cb.x2), fmaxf(kb.x1, cb.x1)), 0.f);
            const float h = fmaxf(__fsub_rn(fminf(kb.y2, cb.y2), fmaxf(kb.y1, cb.y1)), 0.f);
            const float inter = __fmul_rn(w, h);
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(karea, carea), inter));
            if (iou > thr) {
                ++hits;
                const double wt = (double)c.score;
                sw += wt; s0 += wt * c.ob.x1; s1 += wt * c.ob.y1; s2 += wt * c.ob.x2; s3 += wt * c.ob.y2;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            sw += __shfl_xor(sw, d); s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); s3 += __shfl_xor(s3, d);
            hits += __shfl_xor(hits, d);
        }
        if (lane == 0) {
            const float den = (float)sw;
            float* o = rows + k * 6;
            o[0] = __fdiv_rn((float)s0, den); o[1] = __fdiv_rn((float)s1, den); o[2] = __fdiv_rn((float)s2, den); o[3] = __fdiv_rn((float)s3, den);
            o[4] = r[4]; o[5] = r[5];
            rflat[k] = a.kept_index ? a.kept_index[(long long)b * a.max_det + k] : 0;
            alive[k] = hits > 1;
        }
    }
    __syncthreads();
    if (tid == 0) {  // positions of the surviving rows, in order
        int pos = 0;
        for (int k = 0; k < K; ++k) {
            const int keep = alive[k];
            alive[k] = keep ? pos : -1;
            pos += keep;
        }
        newcount = pos;
    }
    __syncthreads();
    for (int k = tid; k < K; k += MERGE_THREADS) {
        const int pos = alive[k];
        if (pos < 0) continue;
        float* o = det + pos * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) o[e] = rows[k * 6 + e];
        if (a.kept_index) a.kept_index[(long long)b * a.max_det + pos] = rflat[k];
    }
    if (tid == 0) a.count[b] = newcount;
}

// zero the unused tail of det rows so that padded outputs are deterministic
__global__ __launch_bounds__(256) void nms_clear_tail_kernel(float* det, int* kept_index, const int* count, int max_det) {
    const int b = blockIdx.x;
    const int c = count[b];
    for (int i = c * 6 + threadIdx.x; i < max_det * 6; i += 256) det[(long long)b * max_det * 6 + i] = 0.f;
    if (kept_index)
        for (int i = c + threadIdx.x; i < max_det; i += 256) kept_index[(long long)b * max_det + i] = -1;
}

// scale_coords + clip_coords (utils/utils.py:193-214) on the padded detection block, in place: one thread per row.
// params[b] = {pad_x, pad_y, gain, img0_w, img0_h}.  Same fp32 operations in the same order as the reference's tensor
// statements: subtract the padding, divide by the gain, clamp.
__global__ __launch_bounds__(256) void scale_coords_kernel(float* __restrict__ det, const int* __restrict__ count,
                                                           const float* __restrict__ params, int max_det, int row_floats, int do_round) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[b] || i >= max_det) return;
    const float px = params[b * 5 + 0], py = params[b * 5 + 1], gain = params[b * 5 + 2];
    const float w0 = params[b * 5 + 3], h0 = params[b * 5 + 4];
    float* r = det + ((long long)b * max_det + i) * row_floats;
    float x1 = __fdiv_rn(__fsub_rn(r[0], px), gain), y1 = __fdiv_rn(__fsub_rn(r[1], py), gain);
    float x2 = __fdiv_rn(__fsub_rn(r[2], px), gain), y2 = __fdiv_rn(__fsub_rn(r[3], py), gain);
    x1 = fminf(fmaxf(x1, 0.f), w0); y1 = fminf(fmaxf(y1, 0.f), h0);
    x2 = fminf(fmaxf(x2, 0.f), w0); y2 = fminf(fmaxf(y2, 0.f), h0);
    if (do_round) { x1 = rintf(x1); y1 = rintf(y1); x2 = rintf(x2); y2 = rintf(y2); }
    r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2;
}

// ------------------------------------------------------------------------------------------
// Prediction <-> target matching of the evaluation loop (test.py:199, 220-247), one workgroup per image.
// ------------------------------------------------------------------------------------------
constexpr int MATCH_THREADS = 256;
constexpr int MATCH_MAX_TARGETS = 2048;   // labels of one image held in LDS
constexpr int MATCH_MAX_IOU = 16;

struct MatchArgs {
    float* det;            // (B, max_det, row_floats): xyxy, conf, cls -- clipped in place like the reference does
    const int* count;      // (B,)
    const float* targets;  // (T, 6): image index, class, x, y, w, h (normalised), the reference's `targets`
    const float* iouv;     // (niou,)
    unsigned char* correct;  // (B, max_det, niou)
    int* overflow;         // (B,) set to 1 when an image has more than MATCH_MAX_TARGETS labels (may be null)
    int n_targets, max_det, row_floats, niou;
    float width, height;
};

__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(const MatchArgs a) {
    __shared__ float tcls[MATCH_MAX_TARGETS];
    __shared__ Box tbox[MATCH_MAX_TARGETS];
    __shared__ unsigned char detected[MATCH_MAX_TARGETS];
    __shared__ float best_iou[MAX_DET_CAP];
    __shared__ int best_t[MAX_DET_CAP];
    __shared__ int nl_s;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = min(a.count[b], a.max_det);
    float* det = a.det + (long long)b * a.max_det * a.row_floats;
    unsigned char* correct = a.correct + (long long)b * a.max_det * a.niou;
    for (int i = tid; i < a.max_det * a.niou; i += MATCH_THREADS) correct[i] = 0;

    // labels of this image, in their original order (the order decides which of two equal IoUs wins): wave 0 compacts
    if (tid < 64) {
        int nl = 0;
        for (int t0 = 0; t0 < a.n_targets; t0 += 64) {
            const int t = t0 + lane;
            const bool mine = t < a.n_targets && a.targets[(long long)t * 6] == (float)b;
            const u64 m = __ballot(mine);
            const int pos = nl + __builtin_popcountll(m & ((1ull << lane) - 1));
            if (mine && pos < MATCH_MAX_TARGETS) {
                const float* r = a.targets + (long long)t * 6;
                // xywh2xyxy(labels[:, 1:5]) * whwh: centre -/+ half size, then scaled to pixels (utils/utils.py:183-190)
                const float hw = r[4] / 2.f, hh = r[5] / 2.f;
                tcls[pos] = r[1];
                tbox[pos] = {__fmul_rn(__fsub_rn(r[2], hw), a.width), __fmul_rn(__fsub_rn(r[3], hh), a.height),
                             __fmul_rn(__fadd_rn(r[2], hw), a.width), __fmul_rn(__fadd_rn(r[3], hh), a.height)};
                detected[pos] = 0;
            }
            nl += __builtin_popcountll(m);
        }
        if (lane == 0) {
            if (a.overflow) a.overflow[b] = nl > MATCH_MAX_TARGETS;
            nl_s = nl < MATCH_MAX_TARGETS ? nl : MATCH_MAX_TARGETS;
        }
    }
    __syncthreads();
    const int nl = nl_s;

    // clip_coords(pred, (height, width)) and, per prediction, the best IoU over the labels of its class (first maximum)
    for (int p = tid; p < n; p += MATCH_THREADS) {
        float* r = det + (long long)p * a.row_floats;
        Box pb = {fminf(fmaxf(r[0], 0.f), a.width), fminf(fmaxf(r[1], 0.f), a.height), fminf(fmaxf(r[2], 0.f), a.width),
                  fminf(fmaxf(r[3], 0.f), a.height)};
        r[0] = pb.x1; r[1] = pb.y1; r[2] = pb.x2; r[3] = pb.y2;
        const float pc = r[5];
        const float parea = __fmul_rn(__fsub_rn(pb.x2, pb.x1), __fsub_rn(pb.y2, pb.y1));
        float best = -1.f;
        int bt = -1;
        for (int t = 0; t < nl; ++t) {
            if (tcls[t] != pc) continue;
            const Box tb = tbox[t];
            // box_iou (utils/utils.py:355-377): clamped width * clamped height over the union, all fp32
            const float w = fmaxf(__fsub_rn(fminf(pb.x2, tb.x2), fmaxf(pb.x1, tb.x1)), 0.f);
            const float h = fmaxf(__fsub_rn(fminf(pb.y2, tb.y2), fmaxf(pb.y1, tb.y1)), 0.f);
            const float inter = __fmul_rn(w, h);
            const float tarea = __fmul_rn(__fsub_rn(tb.x2, tb.x1), __fsub_rn(tb.y2, tb.y1));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(parea, tarea), inter));
            if (iou > best) { best = iou; bt = t; }
        }
        best_iou[p] = best;
        best_t[p] = bt;
    }
    __syncthreads();

    // predictions claim labels in their (descending-confidence) order; a label is claimed once
    if (tid == 0) {
        const float thr0 = a.iouv[0];
        int found = 0;
        for (int p = 0; p < n && found < nl; ++p) {
            const int t = best_t[p];
            if (t < 0 || !(best_iou[p] > thr0) || detected[t]) continue;
            detected[t] = 1;
            ++found;
            for (int k = 0; k < a.niou; ++k) correct[p * a.niou + k] = best_iou[p] > a.iouv[k];
        }
    }
}

int pow2_at_least(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

int ensure_sort_lds() {
    static std::atomic<unsigned long long> done{0};  // one bit per device: the attribute is per function and device
    const unsigned long long done_bit = 1ull << y5_current_device();
    if (!(done.load() & done_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   SORT_LDS_KEYS * (int)sizeof(u64)));
        done.fetch_or(done_bit);
    }
    return Y5_OK;
}

// Sorts every image's keys: the chunked kernels for images of up to CHUNK_MAX_P2 padded keys, the one-workgroup kernel
// for larger ones.  Which path an image takes is decided on the device from its counter; kernels of the other path exit
// at once for it.
int launch_sort(u64* keys, int cap_pow2, int cap, const int* counters, int batch, hipStream_t s) {
    if (int rc = ensure_sort_lds()) return rc;
    const int p2 = cap_pow2 < CHUNK_MAX_P2 ? cap_pow2 : CHUNK_MAX_P2;
    const int chunks = (p2 + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(nms_sort_chunks_kernel, dim3((unsigned)chunks, (unsigned)batch), dim3(1024), 0, s, keys, cap_pow2, cap, counters);
    Y5_LAUNCH_CHECK("nms_sort_chunks_kernel");
    for (int k = 2 * CHUNK; k <= p2; k <<= 1) {
        const int levels = __builtin_ctz((unsigned)(k / CHUNK));
        const dim3 ggrid((unsigned)((p2 >> levels) + 255) / 256, (unsigned)batch);
        if (levels == 1) hipLaunchKernelGGL(nms_merge_global_kernel<1>, ggrid, dim3(256), 0, s, keys, cap_pow2, cap, counters, k);
        else if (levels == 2) hipLaunchKernelGGL(nms_merge_global_kernel<2>, ggrid, dim3(256), 0, s, keys, cap_pow2, cap, counters, k);
        else hipLaunchKernelGGL(nms_merge_global_kernel<3>, ggrid, dim3(256), 0, s, keys, cap_pow2, cap, counters, k);
        Y5_LAUNCH_CHECK("nms_merge_global_kernel");
        hipLaunchKernelGGL(nms_merge_chunks_kernel, dim3((unsigned)chunks, (unsigned)batch), dim3(1024), 0, s, keys, cap_pow2, cap, counters, k);
        Y5_LAUNCH_CHECK("nms_merge_chunks_kernel");
    }
    if (cap_pow2 > CHUNK_MAX_P2) {
        hipLaunchKernelGGL(nms_sort_kernel, dim3((unsigned)batch), dim3(SORT_THREADS), SORT_LDS_KEYS * sizeof(u64), s, keys, cap_pow2, cap,
                           counters, 2 * CHUNK_MAX_P2);
        Y5_LAUNCH_CHECK("nms_sort_kernel");
    }
    return Y5_OK;
}

}  // namespace

extern "C" size_t y5_nms_workspace_bytes(int batch, int max_candidates) {
    if (batch <= 0 || max_candidates <= 0) return 0;
    const size_t counters = ((size_t)batch * sizeof(int) + 255) / 256 * 256;
    return counters + (size_t)batch * pow2_at_least(max_candidates) * sizeof(u64);
}

namespace {

// Argument checks shared by the two batched entry points (everything but the workspace).
int nms_check(const y5_nms_desc* d) {
    Y5_REQUIRE(d, "nms: null descriptor");
    Y5_REQUIRE(d->pred && d->det && d->count, "nms: null pointer");
    Y5_REQUIRE(d->pred_dtype >= 0 && d->pred_dtype <= 2, "nms: pred_dtype %d not in {0 fp16, 1 fp32, 2 bf16}", d->pred_dtype);
    Y5_REQUIRE(d->batch > 0 && d->n_rows > 0 && d->no >= 6, "nms: bad shape (%d, %d, %d)", d->batch, d->n_rows, d->no);
    Y5_REQUIRE((long long)d->n_rows * (d->no - 5) < 0x7fffffffLL, "nms: rows*classes overflows the 32-bit candidate index");
    Y5_REQUIRE(d->max_det > 0 && d->max_det <= MAX_DET_CAP, "nms: max_det %d out of range 1..%d", d->max_det, MAX_DET_CAP);
    Y5_REQUIRE(d->max_candidates > 0, "nms: max_candidates must be positive");
    Y5_REQUIRE(d->conf_thres >= 0.f, "nms: conf_thres must be >= 0");
    return Y5_OK;
}

// The three stages on `workspace` (y5_nms_workspace_bytes(batch, max_candidates) bytes, 16-byte aligned): counters, then keys.
int nms_run(const y5_nms_desc* d, void* workspace, hipStream_t s) {
    const size_t counters_bytes = ((size_t)d->batch * sizeof(int) + 255) / 256 * 256;
    int* counters = static_cast<int*>(workspace);
    u64* keys = reinterpret_cast<u64*>(static_cast<unsigned char*>(workspace) + counters_bytes);
    const int cap_pow2 = pow2_at_least(d->max_candidates);

    Y5_HIP(hipMemsetAsync(counters, 0, counters_bytes, s));
    const dim3 cgrid((unsigned)((d->n_rows + CAND_ROWS - 1) / CAND_ROWS), (unsigned)d->batch);
    const size_t tile_bytes = ((size_t)CAND_ROWS * d->no * (d->pred_dtype == 1 ? 4 : 2) + 15) / 16 * 16;
    Y5_REQUIRE(tile_bytes <= 150 * 1024, "nms: %d values per row do not fit the staging tile", d->no);
    static std::atomic<unsigned long long> cand_attr{0};  // one bit per device: the attribute is per function and device
    const unsigned long long cand_attr_bit = 1ull << y5_current_device();
    if (!(cand_attr.load() & cand_attr_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_candidates_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_candidates_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_candidates_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        cand_attr.fetch_or(cand_attr_bit);
    }
    if (d->pred_dtype == 0)
        hipLaunchKernelGGL(nms_candidates_kernel<f16>, cgrid, dim3(CAND_THREADS), tile_bytes, s, static_cast<const f16*>(d->pred),
                           d->n_rows, d->no, d->conf_thres, d->class_keep, keys, cap_pow2, counters);
    else if (d->pred_dtype == 2)
        hipLaunchKernelGGL(nms_candidates_kernel<bf16>, cgrid, dim3(CAND_THREADS), tile_bytes, s, static_cast<const bf16*>(d->pred),
                           d->n_rows, d->no, d->conf_thres, d->class_keep, keys, cap_pow2, counters);
    else
        hipLaunchKernelGGL(nms_candidates_kernel<float>, cgrid, dim3(CAND_THREADS), tile_bytes, s, static_cast<const float*>(d->pred),
                           d->n_rows, d->no, d->conf_thres, d->class_keep, keys, cap_pow2, counters);
    Y5_LAUNCH_CHECK("nms_candidates_kernel");

    if (int rc = launch_sort(keys, cap_pow2, d->max_candidates, counters, d->batch, s)) return rc;

    SelectArgs a;
    a.pred = d->pred; a.keys = keys; a.counters = counters;
    a.det = d->det; a.count = d->count; a.kept_index = d->kept_index; a.n_candidates = d->n_candidates; a.keep64 = nullptr;
    a.iou_thres = d->iou_thres;
    a.n_rows = d->n_rows; a.no = d->no; a.cap = d->max_candidates; a.cap_pow2 = cap_pow2; a.max_det = d->max_det; a.agnostic = d->agnostic & Y5_NMS_AGNOSTIC;
    if (d->pred_dtype == 0)
        hipLaunchKernelGGL((nms_select_kernel<f16, false>), dim3((unsigned)d->batch), dim3(SEL_THREADS), 0, s, a);
    else if (d->pred_dtype == 2)
        hipLaunchKernelGGL((nms_select_kernel<bf16, false>), dim3((unsigned)d->batch), dim3(SEL_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((nms_select_kernel<float, false>), dim3((unsigned)d->batch), dim3(SEL_THREADS), 0, s, a);
    Y5_LAUNCH_CHECK("nms_select_kernel");
    if (d->agnostic & Y5_NMS_MERGE) {
        static std::atomic<unsigned long long> merge_attr{0};
        const unsigned long long bit = 1ull << y5_current_device();
        if (!(merge_attr.load() & bit)) {
            Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_merge_boxes_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_LDS_BYTES));
            Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_merge_boxes_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_LDS_BYTES));
            Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_merge_boxes_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_LDS_BYTES));
            merge_attr.fetch_or(bit);
        }
        if (d->pred_dtype == 0)
            hipLaunchKernelGGL(nms_merge_boxes_kernel<f16>, dim3((unsigned)d->batch), dim3(MERGE_THREADS), MERGE_LDS_BYTES, s, a);
        else if (d->pred_dtype == 2)
            hipLaunchKernelGGL(nms_merge_boxes_kernel<bf16>, dim3((unsigned)d->batch), dim3(MERGE_THREADS), MERGE_LDS_BYTES, s, a);
        else
            hipLaunchKernelGGL(nms_merge_boxes_kernel<float>, dim3((unsigned)d->batch), dim3(MERGE_THREADS), MERGE_LDS_BYTES, s, a);
        Y5_LAUNCH_CHECK("nms_merge_boxes_kernel");
    }
    hipLaunchKernelGGL(nms_clear_tail_kernel, dim3((unsigned)d->batch), dim3(256), 0, s, d->det, d->kept_index, d->count, d->max_det);
    Y5_LAUNCH_CHECK("nms_clear_tail_kernel");
    return Y5_OK;
}

}  // namespace

extern "C" int y5_nms_batched(const y5_nms_desc* d, y5_stream_t stream) {
    if (int rc = nms_check(d)) return rc;
    Y5_REQUIRE(d->workspace, "nms: null pointer");
    const size_t need = y5_nms_workspace_bytes(d->batch, d->max_candidates);
    if (d->workspace_bytes < need) return y5_fail(Y5_EWORKSPACE, "nms: workspace %zu < %zu bytes", d->workspace_bytes, need);
    Y5_REQUIRE(y5_aligned16(d->workspace), "nms: workspace must be 16-byte aligned");
    return nms_run(d, d->workspace, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------
// The same pipeline for callers that cannot allocate: counters and sort keys live in a device array of this code object.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[Y5_NMS_NOALLOC_BYTES];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
std::atomic<void*> scratch_ptr[NOALLOC_MAX_DEVICES];  // device address of `scratch` on each device, resolved on first use there
}  // namespace

extern "C" int y5_nms_batched_noalloc(const y5_nms_desc* d, y5_stream_t stream) {
    if (int rc = nms_check(d)) return rc;
    const size_t need = y5_nms_workspace_bytes(d->batch, d->max_candidates);
    if (need > (size_t)Y5_NMS_NOALLOC_BYTES)
        return y5_fail(Y5_EWORKSPACE, "nms_noalloc: batch %d x %d candidates needs %zu bytes of scratch, the built-in array holds %zu; "
                       "lower max_candidates or use y5_nms_batched with a workspace", d->batch, d->max_candidates, need, (size_t)Y5_NMS_NOALLOC_BYTES);
    int dev = 0;
    Y5_HIP(hipGetDevice(&dev));
    Y5_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "nms_noalloc: device ordinal %d out of range", dev);
    void* ws = scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ws) {
        // every device holds its own copy of the array at its own address; the array's name is only a host-side handle
        Y5_HIP(hipGetSymbolAddress(&ws, HIP_SYMBOL(scratch)));
        Y5_REQUIRE(ws && y5_aligned16(ws), "nms_noalloc: bad scratch address");
        scratch_ptr[dev].store(ws, std::memory_order_release);
    }
    return nms_run(d, ws, static_cast<hipStream_t>(stream));
}

extern "C" int y5_scale_coords_f32(float* det, const int32_t* count, const float* params, int batch, int max_det, int row_floats,
                                   int do_round, y5_stream_t stream) {
    Y5_REQUIRE(det && count && params, "scale_coords: null pointer");
    Y5_REQUIRE(batch > 0 && max_det > 0 && row_floats >= 4, "scale_coords: bad shape (%d, %d, %d)", batch, max_det, row_floats);
    hipLaunchKernelGGL(scale_coords_kernel, dim3((unsigned)((max_det + 255) / 256), (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), det, count, params, max_det, row_floats, do_round);
    Y5_LAUNCH_CHECK("scale_coords_kernel");
    return Y5_OK;
}

extern "C" int y5_match_predictions_f32(float* det, const int32_t* count, const float* targets, int n_targets, const float* iouv, int niou,
                                        uint8_t* correct, int32_t* overflow, int batch, int max_det, int row_floats, float height,
                                        float width, y5_stream_t stream) {
    Y5_REQUIRE(det && count && iouv && correct, "match: null pointer");
    Y5_REQUIRE(n_targets >= 0 && (n_targets == 0 || targets), "match: null targets");
    Y5_REQUIRE(batch > 0 && max_det > 0 && max_det <= MAX_DET_CAP && row_floats >= 6, "match: bad shape (%d, %d, %d)", batch, max_det, row_floats);
    Y5_REQUIRE(niou >= 1 && niou <= MATCH_MAX_IOU, "match: niou %d out of range 1..%d", niou, MATCH_MAX_IOU);
    MatchArgs a;
    a.det = det; a.count = count; a.targets = targets; a.iouv = iouv; a.correct = correct; a.overflow = overflow;
    a.n_targets = n_targets; a.max_det = max_det; a.row_floats = row_floats; a.niou = niou;
    a.width = width; a.height = height;
    hipLaunchKernelGGL(match_kernel, dim3((unsigned)batch), dim3(MATCH_THREADS), 0, static_cast<hipStream_t>(stream), a);
    Y5_LAUNCH_CHECK("match_kernel");
    return Y5_OK;
}

extern "C" int y5_nms_boxes_f32(const float* boxes, const float* scores, int n, double iou_thres, int64_t* keep,
                                int32_t* keep_count, int max_keep, void* workspace, size_t workspace_bytes,
                                y5_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    Y5_REQUIRE(keep_count && workspace, "nms_boxes: null pointer");
    Y5_REQUIRE(n >= 0, "nms_boxes: negative n");
    Y5_REQUIRE(max_keep > 0 && max_keep <= MAX_DET_CAP, "nms_boxes: max_keep %d out of range 1..%d", max_keep, MAX_DET_CAP);
    if (n == 0) {
        Y5_HIP(hipMemsetAsync(keep_count, 0, sizeof(int32_t), s));
        return Y5_OK;
    }
    Y5_REQUIRE(boxes && scores && keep, "nms_boxes: null pointer");
    Y5_REQUIRE(y5_aligned16(boxes) && y5_aligned16(workspace), "nms_boxes: boxes/workspace must be 16-byte aligned");
    const size_t need = y5_nms_workspace_bytes(1, n);
    if (workspace_bytes < need) return y5_fail(Y5_EWORKSPACE, "nms_boxes: workspace %zu < %zu bytes", workspace_bytes, need);
    int* counters = static_cast<int*>(workspace);
    u64* keys = reinterpret_cast<u64*>(static_cast<unsigned char*>(workspace) + 256);
    const int cap_pow2 = pow2_at_least(n);
    hipLaunchKernelGGL(nms_box_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, n, keys, counters);
    Y5_LAUNCH_CHECK("nms_box_keys_kernel");
    if (int rc = launch_sort(keys, cap_pow2, n, counters, 1, s)) return rc;
    SelectArgs a;
    a.pred = boxes; a.keys = keys; a.counters = counters;
    a.det = nullptr; a.count = keep_count; a.kept_index = nullptr; a.n_candidates = nullptr; a.keep64 = reinterpret_cast<long long*>(keep);
    a.iou_thres = iou_thres;
    a.n_rows = n; a.no = 6; a.cap = n; a.cap_pow2 = cap_pow2; a.max_det = max_keep; a.agnostic = 1;
    hipLaunchKernelGGL((nms_select_kernel<float, true>), dim3(1), dim3(SEL_THREADS), 0, s, a);
    Y5_LAUNCH_CHECK("nms_select_kernel");
    return Y5_OK;
}
