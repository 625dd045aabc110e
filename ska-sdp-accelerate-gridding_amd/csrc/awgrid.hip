// AW-projection gridders: convgrid3 / convgrid4 (src/Gridding.hs:246-396), which produce the same
// grid, and aw_kernel_fn2 / convolve2d (:761-811) that build each visibility's kernel
//     awkern_k = conj( convolve2d( convolve2d(akerns[a1], akerns[a2]), wkerns[wbin, yf, xf] ) ).
//
// The reference evaluates convolve2d with six 32x32 FFTs per visibility inside a sequential
// `awhile`.  Its pad_mid transposes the operands (padder reads `array ! index2 oldx oldy`, :875),
// so convolve2d(a, b) = same_conv(a, b)^T exactly; here that is evaluated directly:
//   1. the antenna-pair product convolve2d(a1, a2) is computed once per pair that occurs;
//   2. visibilities are de-duplicated by KEY = (pair, wbin, yf, xf) with a device hash table (option
//      "aw_cache"; real data repeats a key over the consecutive dumps of a baseline): the kernel of every
//      distinct key is built once, on the fp64 vector ALU (aw_build_kernel);
//   3. the table of distinct kernels feeds the tap-reusing tile gridder (tile_sorted.hip, AW mode): a record's
//      slice is its key's index, runs of equal key reuse their taps from registers.
// Work is done in batches so the kernel table stays bounded; nothing is read back inside a call.
#include "tile_common.h"

namespace gridhip {

constexpr uint64_t AW_EMPTY = ~0ull;

// out[x*S + y] = sum_{i,j} a[i][j] * b[y-i+c][x-j+c]   (c = S/2; a, b in LDS)   == same_conv(a,b)^T
__device__ __forceinline__ void conv_same_T(const double2 *a, const double2 *b, int S, double2 *out, bool conj)
{
    const int c = S / 2;
    for (int o = threadIdx.x; o < S * S; o += blockDim.x) {
        const int x = o / S, y = o - x * S;
        double sr = 0.0, si = 0.0;
        const int ilo = max(0, y + c - (S - 1)), ihi = min(S - 1, y + c);
        const int jlo = max(0, x + c - (S - 1)), jhi = min(S - 1, x + c);
        for (int i = ilo; i <= ihi; ++i) {
            const double2 *arow = a + i * S;
            const double2 *brow = b + (y - i + c) * S + (x + c);
            for (int j = jlo; j <= jhi; ++j) {
                const double2 av = arow[j], bv = brow[-j];
                sr += av.x * bv.x - av.y * bv.y;
                si += av.x * bv.y + av.y * bv.x;
            }
        }
        out[o] = make_double2(sr, conj ? -si : si);
    }
}

// ---- 1. antenna pairs that occur -> dense slots (first come first served; nothing depends on the order)
//   slot[p*A+q]: -1 unseen; counters[0] = pairs so far; pairlist[slot] = p*A+q
__global__ void aw_pairs_kernel(int64_t n, int64_t A, const int64_t *__restrict__ a1, const int64_t *__restrict__ a2,
                                int32_t *__restrict__ slot, int32_t *__restrict__ pairlist, int32_t *__restrict__ counters,
                                int32_t cap)
{
    const int lane = threadIdx.x & 63;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = a1[k], q = a2[k];
        int32_t *s = nullptr;
        bool won = false;
        const bool valid = p >= 0 && p < A && q >= 0 && q < A;
        // consecutive samples of a baseline sit in neighbouring lanes: only the first lane of such a run tries the claim
        const int64_t pq = valid ? p * A + q : -1;
        const int64_t prev = __shfl_up(pq, 1);
        if (valid && (lane == 0 || prev != pq)) {
            s = slot + pq;
            // (claimed already: nothing to do; a stale -1 only costs the atomic)
            if (*s == -1) won = atomicCAS(s, -1, -2) == -1;
        }
        // the wave's new pairs are numbered together: one atomic on the counter per wave, not per pair (10^5 atomics
        // on one address were most of this kernel's 0.19 ms)
        const unsigned long long m = __ballot(won);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&counters[0], __popcll(m));
            base = __shfl(base, leader);
            if (won) {
                const int32_t id = base + __popcll(m & ((1ull << lane) - 1));
                if (id < cap) pairlist[id] = (int32_t)(p * A + q);
                __hip_atomic_store(s, id < cap ? id : -3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// pairk[slot] = convolve2d(akerns[a1], akerns[a2]) for every pair that occurs (one work-group per pair at a time)
__global__ void aw_pair_kernel(int64_t A, int S, const double2 *__restrict__ akerns, const int32_t *__restrict__ pairlist,
                               const int32_t *__restrict__ counters, int32_t cap, double2 *__restrict__ pairk,
                               int32_t *__restrict__ errors)
{
    extern __shared__ double2 sm[];
    const int npairs = min(counters[0], cap);
    double2 *la = sm, *lb = sm + S * S;
    for (int s = blockIdx.x; s < npairs; s += gridDim.x) {
        int64_t pq = pairlist[s];
        if (pq < 0 || pq >= A * A) {  // (cannot happen: aw_pairs_kernel wrote it)
            pq = 0;
            if (threadIdx.x == 0) atomicAdd(errors, 1);
        }
        const int64_t p = pq / A, q = pq - p * A;
        __syncthreads();
        for (int t = threadIdx.x; t < S * S; t += blockDim.x) {
            la[t] = akerns[p * S * S + t];
            lb[t] = akerns[q * S * S + t];
        }
        __syncthreads();
        conv_same_T(la, lb, S, pairk + (size_t)s * S * S, false);
    }
}

// ---- 2. keys.  key = pair slot << 30 | w-kernel slice (wslice < 2^30, pair slot < 2^31).
// cache: insert into an open-addressing table (64-bit CAS); the first to insert a key numbers it.
//   kid[k]  : index of the visibility's kernel in this batch's table, or -1 (dropped: index out of range)
//   ok[k]   : 0 / -1, handed to the binning pre-pass as the "w-plane" of a one-plane table (-1 is dropped and counted)
//   ukey[id]: the key of table entry id;  counters[1] = entries so far
__global__ void aw_keys_kernel(int64_t H, int64_t Wd, int64_t n, int64_t W, int32_t Q, int64_t A,
                               const double *__restrict__ u, const double *__restrict__ v, int64_t stride,
                               const int64_t *__restrict__ wbin, const int64_t *__restrict__ a1,
                               const int64_t *__restrict__ a2, const int32_t *__restrict__ slot, int cache,
                               unsigned long long *__restrict__ htab, int32_t *__restrict__ hid, uint32_t hmask,
                               int32_t *__restrict__ kid, int64_t *__restrict__ ok, unsigned long long *__restrict__ ukey,
                               int32_t *__restrict__ counters, int32_t cap)
{
    const int lane = threadIdx.x & 63;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wb = wbin[k], p = a1[k], q = a2[k];
        const double pu = u[k * stride], pv = v[k * stride];
        int32_t ps = -1;
        if (p >= 0 && p < A && q >= 0 && q < A) ps = slot[p * A + q];
        // the reference would index out of range; such a visibility contributes nothing and is counted
        const bool bad = wb < 0 || wb >= W || ps < 0 || !(pu == pu) || !(pv == pv);
        int32_t id = -1;
        bool won = false;
        uint32_t hwon = 0;
        unsigned long long kwon = 0;
        unsigned long long key = AW_EMPTY;
        if (!bad) {
            int64_t x, y;
            int32_t xf, yf;
            frac_coord_dev(Wd, Q, pu, &x, &xf);
            frac_coord_dev(H, Q, pv, &y, &yf);
            key = ((unsigned long long)ps << 30) | (unsigned long long)((wb * Q + yf) * Q + xf);
            if (!cache) {
                id = (int32_t)k;
                ukey[k] = key;
            }
        }
        if (cache) {
            // consecutive samples of a baseline mostly share their key and sit in neighbouring lanes: only the first lane
            // of such a run probes the table, the others take its slot (the atomics of this kernel are what it costs)
            const unsigned long long prevkey = __shfl_up(key, 1);
            const bool first = !bad && (lane == 0 || prevkey != key);
            uint32_t h = 0;
            if (first) {
                h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & hmask;
                for (;;) {
                    const unsigned long long cur = __hip_atomic_load(&htab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == key) break;
                    if (cur == AW_EMPTY) {
                        const unsigned long long old = atomicCAS(&htab[h], AW_EMPTY, key);
                        if (old == AW_EMPTY) {  // ours: numbered below, together with the wave's other new keys
                            won = true;
                            break;
                        }
                        if (old == key) break;
                    }
                    h = (h + 1) & hmask;  // (the table has at least twice as many slots as there are keys)
                }
                hwon = h;
                kwon = key;
            }
            // the run's first lane: the highest lane at or below this one that probed
            const unsigned long long fm = __ballot(first) & ((2ull << lane) - 1);
            const int src = fm ? 63 - __clzll((long long)fm) : lane;
            const uint32_t hrun = (uint32_t)__shfl((int)h, src);
            if (!bad) id = -(int32_t)hrun - 16;  // resolved by aw_kid_kernel once every number has been handed out
        }
        // one atomic on the counter per wave for all of its new keys (3 x 10^5 atomics on one address were most of this
        // kernel's 0.33 ms); nobody waits for hid[] inside this launch
        const unsigned long long m = __ballot(won);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&counters[1], __popcll(m));
            base = __shfl(base, leader);
            if (won) {
                const int32_t nid = base + __popcll(m & ((1ull << lane) - 1));
                if (nid < cap) ukey[nid] = kwon;
                __hip_atomic_store(&hid[hwon], nid < cap ? nid : -3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (bad && !cache) ukey[k] = 0;  // (every table entry is built: give the dropped one's a key that exists)
        kid[k] = id;
        ok[k] = bad ? -1 : 0;
    }
}

// second half of the cached form: hash slot -> number (a separate launch, so that every number is visible)
__global__ void aw_kid_kernel(int64_t n, const int32_t *__restrict__ hid, int32_t *__restrict__ kid, int64_t *__restrict__ ok)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = kid[k];
        if (c <= -16) {
            const int32_t id = hid[-(c + 16)];
            kid[k] = id;
            if (id < 0) ok[k] = -1;  // (cannot happen: the table holds as many kernels as the batch has visibilities)
        }
    }
}

// ---- 3. kernels of the distinct keys: table[id] = conj(convolve2d(pairk[pair], wkerns[wslice]))
// generic support: one work-group per key at a time
// Nothing read from memory becomes an address before it is in range (the rule of the tile kernels, tile_common.h):
// a key names a pair slot below `npair` and a w-kernel slice below `nslice`; anything else (a stale or overwritten
// table entry - round 2 saw one abort the process) is brought into range and counted in *errors.
__device__ __forceinline__ void aw_key_operands(unsigned long long key, int64_t npair, int64_t nslice, size_t *ia, size_t *ib,
                                                int *bad)
{
    unsigned long long a = key >> 30, b = key & 0x3fffffffull;
    if (a >= (unsigned long long)npair) {
        a = 0;
        ++*bad;
    }
    if (b >= (unsigned long long)nslice) {
        b = 0;
        ++*bad;
    }
    *ia = (size_t)a;
    *ib = (size_t)b;
}

__global__ void aw_build_generic_kernel(int S, const double2 *__restrict__ wkerns, const double2 *__restrict__ pairk,
                                        const unsigned long long *__restrict__ ukey, const int32_t *__restrict__ counters,
                                        int which, int32_t fixed_count, int32_t cap, double2 *__restrict__ table,
                                        int64_t npair_cap, int64_t nslice, int32_t *__restrict__ errors)
{
    extern __shared__ double2 sm[];
    const int nk = min(which >= 0 ? counters[which] : fixed_count, cap);
    const int64_t npair = min((int64_t)counters[0], npair_cap);
    double2 *la = sm, *lb = sm + S * S;
    for (int id = blockIdx.x; id < nk; id += gridDim.x) {
        size_t ia, ib;
        int bad = 0;
        aw_key_operands(ukey[id], npair, nslice, &ia, &ib, &bad);
        if (bad && threadIdx.x == 0) atomicAdd(errors, bad);
        const double2 *pk = pairk + ia * S * S;
        const double2 *wk = wkerns + ib * S * S;
        __syncthreads();
        for (int t = threadIdx.x; t < S * S; t += blockDim.x) {
            la[t] = pk[t];
            lb[t] = wk[t];
        }
        __syncthreads();
        conv_same_T(la, lb, S, table + (size_t)id * S * S, true);
    }
}

// The staged operands are native two-element vectors, not double2 structs: assigning a struct from global memory is a
// memcpy into the array, and an array that is the target of memcpys and lives across the group loop is left in scratch
// memory by the compiler (seen: 60 scratch stores and 30 scratch loads per group).
typedef double aw_d2 __attribute__((ext_vector_type(2)));

// Requests the operands of kernel id_ of a builder group into sa / sb: which S x S arrays it convolves is brought into
// range before it becomes an address; all of the lane's loads together (a load-store loop would pay the memory latency
// per trip).
template <int S, bool PAIR, int NE>
__device__ __forceinline__ void aw_request_operands(aw_d2 (&sa)[NE], aw_d2 (&sb)[NE], int64_t id_, int nk, int x,
                                                    const double2 *__restrict__ wkerns, const double2 *__restrict__ pairk,
                                                    const unsigned long long *__restrict__ ukey,
                                                    const int32_t *__restrict__ pairlist, int64_t A, int64_t npair,
                                                    int64_t nslice, int32_t *__restrict__ errors)
{
    constexpr int S2 = S * S;
    size_t ia = 0, ib = 0;
    if (id_ < nk) {
        int bad = 0;
        if (PAIR) {
            int64_t pq = pairlist[id_];
            if (pq < 0 || pq >= A * A) {
                pq = 0;
                ++bad;
            }
            ia = (size_t)(pq / A);
            ib = (size_t)(pq - (int64_t)ia * A);
        } else
            aw_key_operands(ukey[id_], npair, nslice, &ia, &ib, &bad);
        if (bad && x == 0) atomicAdd(errors, bad);
    }
    const aw_d2 *pk = reinterpret_cast<const aw_d2 *>(pairk + ia * S2);
    const aw_d2 *wk = reinterpret_cast<const aw_d2 *>(wkerns + ib * S2);
#pragma unroll
    for (int t = 0; t < NE; ++t) {
        const int e = min(x + 16 * t, S2 - 1);
        sa[t] = pk[e];
        sb[t] = wk[e];
    }
}

// Compile-time support, organised for the fp64 vector ALU.  A wave builds four kernels at a time: lane = (kernel q,
// column x), and the lane keeps out[y][x] for all S rows y in registers.  The loop nest is ordered so that one LDS
// read feeds many FMAs: for each column j of the pair kernel the lane fetches the S values a[i][j] (all i) and the
// S values b[r][x + c - j] (all r) - 2 S reads - and then performs the S^2 - c(c+1) products of the pairs (i, r)
// whose output row y = r + i - c exists, four FMAs each, from registers: 30 reads per 169 complex products at
// 15 x 15.  b comes from an LDS copy of the w-kernel slice whose rows are followed by c zeros (row pitch S + c, c
// zeros in front of row 0), so a column index of -c .. S-1+c needs no bounds logic.
// PAIR: the same machine builds the antenna-pair products: entry s = convolve2d(akerns[p], akerns[q]) for
// pairlist[s] = p * A + q (wkerns = pairk = akerns, not conjugated).
// ABL (tuning builds, option "dbg"): 1 = the operands are read from LDS once per group instead of once per column j
// (wrong results: what the loop costs without its LDS reads), 2 = no stores of the results, 4 = the group's operands
// requested at its start, not a group ahead (right results)
template <int S, bool PAIR, int ABL = 0>
__global__ void __launch_bounds__(256) aw_build_kernel(const double2 *__restrict__ wkerns, const double2 *__restrict__ pairk,
                                                       const unsigned long long *__restrict__ ukey,
                                                       const int32_t *__restrict__ pairlist, int64_t A,
                                                       const int32_t *__restrict__ counters, int which, int32_t fixed_count,
                                                       int32_t cap, double2 *__restrict__ table, int64_t npair_cap,
                                                       int64_t nslice, int32_t *__restrict__ errors)
{
    constexpr int C = S / 2, PB = S + C, S2 = S * S, BSZ = C + S * PB;  // padded slice: C zeros, then rows of S values + C zeros
    static_assert(S <= 16, "one 16-lane row per kernel");
    extern __shared__ double2 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // shader clock held during this launch (read-only option "aw_clock_khz"): cycle counter and 100 MHz real-time counter
    // stamped by the first work-group when it starts and when it is done; errors points at scalars[31], the stamps are
    // scalars[96..103]
    long long *stamp = reinterpret_cast<long long *>(errors + (96 - 31));
    if (!PAIR && blockIdx.x == 0 && threadIdx.x == 0) {
        stamp[0] = (long long)__builtin_amdgcn_s_memtime();
        stamp[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    const int q = lane >> 4, x = lane & 15;
    double2 *abuf = sm + (size_t)(wave * 4 + q) * (S2 + BSZ);  // this kernel's pair kernel ...
    double2 *bbuf = abuf + S2;                                  // ... and padded w-kernel slice
    const int nk = min(which >= 0 ? counters[which] : fixed_count, cap);
    // the zeros of the padded slices are written once
    for (int e = x; e < BSZ; e += 16) {
        const int pos = e - C, col = pos >= 0 ? pos % PB : -1;
        if (pos < 0 || col >= S) bbuf[e] = make_double2(0.0, 0.0);
    }
    const int64_t groups = ((int64_t)nk + 3) / 4, stride = (int64_t)gridDim.x * 4;
    constexpr int NE = (S2 + 15) / 16;
    constexpr bool AHEAD = !(ABL & 4);  // the next group's operands are requested before the current group's products
    aw_d2 sa[NE], sb[NE];
    const int64_t npair = PAIR ? 0 : min((int64_t)counters[0], npair_cap);
#define GH_REQUEST(g_) aw_request_operands<S, PAIR, NE>(sa, sb, (g_) * 4 + q, nk, x, wkerns, pairk, ukey, pairlist, A, npair, nslice, errors)
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    if (AHEAD && grp < groups) GH_REQUEST(grp);
    for (; grp < groups; grp += stride) {
        const int64_t id = grp * 4 + q;
        const bool have = id < nk;
        if (!AHEAD) GH_REQUEST(grp);
        __builtin_amdgcn_wave_barrier();  // (the previous group's reads of this LDS region are done: same wave)
#pragma unroll
        for (int t = 0; t < NE; ++t) {
            const int e = x + 16 * t;
            if (e < S2) {
                *reinterpret_cast<aw_d2 *>(abuf + e) = sa[t];
                *reinterpret_cast<aw_d2 *>(bbuf + C + (e / S) * PB + (e % S)) = sb[t];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // With one wave per SIMD nothing else covers the ~2 us of these loads: they are in flight during the products
        // (120 registers at 15 x 15, which the one wave has: 512 per lane).
        if (AHEAD) GH_REQUEST(grp + stride);  // (past the last group: kernel 0's operands, not used)

        double2 acc[S];
#pragma unroll
        for (int y = 0; y < S; ++y) acc[y] = make_double2(0.0, 0.0);
        const int xc = x < S ? x : S - 1;  // (the 16th lane of a row repeats the 15th: its results are not stored)
        double2 acol[S], bcol[S];
        if (ABL & 1) {
#pragma unroll
            for (int i = 0; i < S; ++i) acol[i] = abuf[i * S];
#pragma unroll
            for (int r = 0; r < S; ++r) bcol[r] = bbuf[C + xc + C + r * PB];
        }
        for (int j = 0; j < S; ++j) {
            const double2 *bp = bbuf + C + (xc + C - j);  // b[r][xc + C - j] = bp[r * PB]
            if (!(ABL & 1)) {
#pragma unroll
                for (int i = 0; i < S; ++i) acol[i] = abuf[i * S + j];
#pragma unroll
                for (int r = 0; r < S; ++r) bcol[r] = bp[r * PB];
            } else
                asm volatile("" : "+v"(acol[0].x), "+v"(bcol[0].x));  // (keeps the loop a loop)
#pragma unroll
            for (int i = 0; i < S; ++i) {
#pragma unroll
                for (int r = 0; r < S; ++r) {
                    const int y = r + i - C;
                    if (y < 0 || y >= S) continue;  // (compile time)
                    acc[y].x = fma(acol[i].x, bcol[r].x, acc[y].x);
                    acc[y].x = fma(-acol[i].y, bcol[r].y, acc[y].x);
                    acc[y].y = fma(acol[i].x, bcol[r].y, acc[y].y);
                    acc[y].y = fma(acol[i].y, bcol[r].x, acc[y].y);
                }
            }
        }
        if (have && x < S && !((ABL & 2) && acc[0].x != 1.2345e300)) {  // out[x * S + y]: the result comes out transposed, and is conjugated
            double2 *out = table + (size_t)id * S2 + (size_t)x * S;
#pragma unroll
            for (int y = 0; y < S; ++y) out[y] = make_double2(acc[y].x, PAIR ? acc[y].y : -acc[y].y);
        }
    }
    if (!PAIR && blockIdx.x == 0 && threadIdx.x == 0) {
        stamp[2] = (long long)__builtin_amdgcn_s_memtime();
        stamp[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef GH_REQUEST
}

// records come out of the binning pre-pass with kslice = the visibility's index: replace it by its kernel's
__global__ void aw_relabel_kernel(Geom g, RecWord *__restrict__ recs, const int32_t *__restrict__ nrec,
                                  const int32_t *__restrict__ kid, int32_t nvis)
{
    const int n = min(*nrec, g.nrec);
    const RecWord kmask = ((1ull << g.kb) - 1) << g.ob;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RecWord w = recs[i];
        const uint32_t o = (uint32_t)(w & ((1ull << g.ob) - 1));
        const uint32_t k = o < (uint32_t)nvis ? (uint32_t)max(kid[o], 0) : 0u;
        recs[i] = (w & ~kmask) | ((RecWord)k << g.ob & kmask);
    }
}

// [30] += [0]: the drops of one batch's binning join the call's total, [31] += [2] its internal errors (every batch's
// binning clears [0..2]; the kernel builders count theirs in [31] directly); [28] += the batch's distinct kernels
__global__ void aw_account_kernel(int32_t *__restrict__ scalars, const int32_t *__restrict__ counters, int cache, int32_t m)
{
    scalars[30] += scalars[0];
    scalars[31] += scalars[2];
    scalars[28] += cache ? counters[1] : m;
    scalars[29] += m;
}
__global__ void aw_finish_kernel(int32_t *__restrict__ scalars)
{
    scalars[0] = scalars[30];
    scalars[2] = scalars[31];
}

}  // namespace gridhip

using namespace gridhip;

namespace {
inline bool cache_on(const gridhip_ctx *ctx) { return ctx->opt.aw_cache != 0; }
template <int S, bool PAIR>
int launch_build(gridhip_ctx *ctx, const double2 *wk, const double2 *pairk, const unsigned long long *ukey,
                 const int32_t *pairlist, int64_t A, const int32_t *counters, int which, int32_t fixed, int32_t cap,
                 double2 *table, int64_t npair_cap, int64_t nslice)
{
    constexpr int C = S / 2, PB = S + C, BSZ = C + S * PB;
    const size_t lds = (size_t)16 * (S * S + BSZ) * sizeof(double2);
#ifdef GRIDHIP_TUNING
    if (S == 15 && !PAIR && ctx->opt.dbg >= 1 && ctx->opt.dbg <= 4) {  // ablations of the key builder (1 - 3: wrong results)
#define AW_ABL1(A_)                                                                                                         \
    if (ctx->opt.dbg == A_) {                                                                                               \
        GH_CHECK(raise_lds(ctx, aw_build_kernel<15, false, A_>));                                                           \
        hipLaunchKernelGGL((aw_build_kernel<15, false, A_>), dim3(ctx->num_cu), dim3(256), lds, ctx->stream, wk, pairk, ukey, \
                           pairlist, A, counters, which, fixed, cap, table, npair_cap, nslice, ctx->d_scalars + 31);       \
    }
        AW_ABL1(1) AW_ABL1(2) AW_ABL1(3) AW_ABL1(4)
#undef AW_ABL1
        return GRIDHIP_OK;
    }
#endif
    GH_CHECK(raise_lds(ctx, aw_build_kernel<S, PAIR>));
    // two work-groups per CU where the LDS holds them (supports up to 11 x 11): two waves per SIMD cover each other's
    // LDS and memory latencies - build phase 1.61 -> 1.51 ms at 11 x 11, 1.18 -> 1.09 ms at 9 x 9 (10^6 visibilities)
    const int nwg = ctx->num_cu * (2 * lds <= (size_t)ctx->max_lds ? 2 : 1);
    // one work-group (four waves, one per SIMD) per CU; the loop strides over the entries
    hipLaunchKernelGGL((aw_build_kernel<S, PAIR>), dim3(nwg), dim3(256), lds, ctx->stream, wk, pairk, ukey, pairlist, A,
                       counters, which, fixed, cap, table, npair_cap, nslice, ctx->d_scalars + 31);
    return GRIDHIP_OK;
}
}  // namespace

extern "C" {

// Device pointers; asynchronous (nothing is read back; scratch grows on first use).
int gridhip_awgrid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                       int64_t S, int64_t A, const double *wkerns, const double *akerns, const double *u,
                       const double *v, int64_t uv_stride, const int64_t *wbin, const int64_t *a1,
                       const int64_t *a2, const double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || W <= 0 || Q <= 0 || S <= 0 || A <= 0 || uv_stride < 1 || !grid || !wkerns ||
        !akerns || (n > 0 && (!u || !v || !wbin || !a1 || !a2 || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    if (S > 63 || A > 46340 || n > (int64_t)0x7fffff00 || W * Q * Q >= ((int64_t)1 << 30))
        return fail(ctx, GRIDHIP_EUNSUPPORTED, "shape outside aw limits");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    // [0] dropped, [2] errors, [28] distinct kernels built, [29] visibilities keyed, [30] dropped so far (batches),
    // [31] errors so far
    GH_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(int32_t), ctx->stream));
    GH_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 28, 0, 4 * sizeof(int32_t), ctx->stream));
    if (n == 0) return GRIDHIP_OK;
    const size_t S2 = (size_t)S * S, pairs = (size_t)A * A;
    const int cache = ctx->opt.aw_cache != 0;
    // The table of kernels is sized for the batch whatever the de-duplication finds (nothing is read back inside a
    // call): 2^20 visibilities per batch with the cache (3.8 GB of table at 15 x 15; the 10^6-visibility benchmark is
    // one batch), 2^22 without it, where every visibility has a kernel of its own anyway (15 GB).
    const int64_t bcap = cache_on(ctx) ? ((int64_t)1 << 20) : ((int64_t)1 << 22);
    const int64_t batch = n < bcap ? n : bcap;
    const int32_t pair_cap = (int32_t)(pairs < (size_t)n ? pairs : (size_t)n);
    uint32_t hslots = 1024;
    while (hslots < 2 * (uint64_t)batch) hslots <<= 1;

    // ---- scratch (grows on first use; laid out in one workspace)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_slot = 0, o_plist = o_slot + al(pairs * 4), o_cnt = o_plist + al((size_t)pair_cap * 4),
                 o_pairk = o_cnt + al(64), o_htab = o_pairk + al((size_t)pair_cap * S2 * 16),
                 o_hid = o_htab + al((size_t)hslots * 8), o_kid = o_hid + al((size_t)hslots * 4),
                 o_ok = o_kid + al((size_t)batch * 4), o_ukey = o_ok + al((size_t)batch * 8),
                 o_table = o_ukey + al((size_t)batch * 8), total = o_table + al((size_t)batch * S2 * 16);
    GH_CHECK(ws_reserve(ctx, ctx->aw, total));
    char *base = (char *)ctx->aw.ptr;
    int32_t *slot = (int32_t *)(base + o_slot), *pairlist = (int32_t *)(base + o_plist), *counters = (int32_t *)(base + o_cnt);
    double2 *pairk = (double2 *)(base + o_pairk), *table = (double2 *)(base + o_table);
    unsigned long long *htab = (unsigned long long *)(base + o_htab), *ukey = (unsigned long long *)(base + o_ukey);
    int32_t *hid = (int32_t *)(base + o_hid), *kid = (int32_t *)(base + o_kid);
    int64_t *ok = (int64_t *)(base + o_ok);

    // geometry of the gridding step (one plane: `ok` stands in for wbin; every slice is private to its key)
    Prep p;
    {
        const int64_t keep_w = ctx->opt.wgroups;
        ctx->opt.wgroups = 1;  // the table is read once per run: nothing for an L2 to keep
        Geom g;
        int block;
        size_t lds;
        int rc = make_geom(ctx, H, Wd, 1, Q, S, S, batch, &g, &block, &lds, 0);  // (no big-tile form of the aw tile kernel)
        ctx->opt.wgroups = keep_w;
        GH_CHECK(rc);
        g.per_vis = 1;
        g.nslices = (int32_t)batch;  // table capacity: a record's kslice is brought below it
        set_rec_bits(&g);
        p.g = g;
        p.block = block;
        p.lds = lds;
        p.nrec = batch;
        p.sorted = ctx->opt.sort != 2 && sorted_plan(ctx, p.g, p.block, &p.nkeys, &p.batch, &p.lds_sorted);
        if (p.sorted) p.g.chunk = p.batch;
    }
    GH_CHECK(ws_reserve(ctx, ctx->tables, tables_bytes(p.g)));
    GH_CHECK(ws_reserve(ctx, ctx->recs, (size_t)batch * sizeof(RecWord)));

    mark(ctx, 0);
    // ---- antenna-pair kernels
    GH_CHECK_HIP(ctx, hipMemsetAsync(slot, 0xff, pairs * 4, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemsetAsync(counters, 0, 64, ctx->stream));
    int64_t blocks = (n + 255) / 256;
    if (blocks > ctx->num_cu * 8) blocks = ctx->num_cu * 8;
    hipLaunchKernelGGL(aw_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n, A, a1, a2, slot, pairlist,
                       counters, pair_cap);
    switch (S) {
#define AW_CASE(S_) \
    case S_: GH_CHECK((launch_build<S_, true>(ctx, (const double2 *)akerns, (const double2 *)akerns, nullptr, pairlist, A, counters, 0, 0, pair_cap, pairk, (int64_t)pair_cap, (int64_t)A))); break;
        AW_CASE(5) AW_CASE(7) AW_CASE(9) AW_CASE(11) AW_CASE(13) AW_CASE(15)
#undef AW_CASE
        default: {
            int pb = pair_cap < ctx->num_cu * 8 ? pair_cap : ctx->num_cu * 8;
            hipLaunchKernelGGL(aw_pair_kernel, dim3((unsigned)(pb > 0 ? pb : 1)), dim3(256), 2 * S2 * sizeof(double2), ctx->stream,
                               A, (int)S, (const double2 *)akerns, pairlist, counters, pair_cap, pairk, ctx->d_scalars + 31);
        }
    }
    GH_CHECK_HIP(ctx, hipGetLastError());

    // ---- per batch: keys -> distinct kernels -> binning -> tile gridder
    for (int64_t lo = 0; lo < n; lo += batch) {
        const int64_t m = n - lo < batch ? n - lo : batch;
        const double *bu = u + lo * uv_stride, *bv = v + lo * uv_stride;
        if (cache) {
            GH_CHECK_HIP(ctx, hipMemsetAsync(htab, 0xff, (size_t)hslots * 8, ctx->stream));
            GH_CHECK_HIP(ctx, hipMemsetAsync(counters + 1, 0, 4, ctx->stream));
        }
        int64_t kb = (m + 255) / 256;
        if (kb > ctx->num_cu * 8) kb = ctx->num_cu * 8;
        hipLaunchKernelGGL(aw_keys_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, H, Wd, m, W, (int32_t)Q, A, bu, bv,
                           uv_stride, wbin + lo, a1 + lo, a2 + lo, slot, cache, htab, hid, hslots - 1, kid, ok, ukey,
                           counters, (int32_t)batch);
        if (cache)
            hipLaunchKernelGGL(aw_kid_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, m, hid, kid, ok);
        const int which = cache ? 1 : -1;
        switch (S) {
#define AW_CASE(S_) \
    case S_: GH_CHECK((launch_build<S_, false>(ctx, (const double2 *)wkerns, pairk, ukey, nullptr, A, counters, which, (int32_t)m, (int32_t)batch, table, (int64_t)pair_cap, (int64_t)(W * Q * Q)))); break;
            AW_CASE(5) AW_CASE(7) AW_CASE(9) AW_CASE(11) AW_CASE(13) AW_CASE(15)
#undef AW_CASE
            default: {
                int gb = (int)(m < ctx->num_cu * 8 ? m : ctx->num_cu * 8);
                hipLaunchKernelGGL(aw_build_generic_kernel, dim3((unsigned)gb), dim3(256), 2 * S2 * sizeof(double2), ctx->stream,
                                   (int)S, (const double2 *)wkerns, pairk, ukey, counters, which, (int32_t)m, (int32_t)batch,
                                   table, (int64_t)pair_cap, (int64_t)(W * Q * Q), ctx->d_scalars + 31);
            }
        }
        GH_CHECK_HIP(ctx, hipGetLastError());
        mark(ctx, 1);
        // gridding: the pre-pass drops the visibilities whose `ok` is -1 (counted), the tile kernel reads the table
        p.g.nvis = (int32_t)m;
        p.g.nrec = (int32_t)m;
        GH_CHECK(launch_bin(ctx, p.g, m, bu, bv, uv_stride, ok));
        Tables t = tables_of(ctx, p.g);
        {
            int rb = (int)((m + 255) / 256);
            if (rb > ctx->num_cu * 8) rb = ctx->num_cu * 8;
            hipLaunchKernelGGL(aw_relabel_kernel, dim3((unsigned)rb), dim3(256), 0, ctx->stream, p.g, (RecWord *)ctx->recs.ptr,
                               t.bin_start + p.g.nbins, kid, (int32_t)m);
        }
        if (p.sorted)
            GH_CHECK(launch_tile_grid_sorted(ctx, p.g, p.block, p.lds_sorted, p.nkeys, p.batch, m, (const double *)table,
                                             vis + 2 * lo, grid, false));
        else
            GH_CHECK(launch_tile_grid(ctx, p.g, p.block, p.lds, m, (const double *)table, vis + 2 * lo, grid));
        hipLaunchKernelGGL(aw_account_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scalars, counters, cache, (int32_t)m);
        mark(ctx, 2);
    }
    hipLaunchKernelGGL(aw_finish_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scalars);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

// What the last gridhip_awgrid*_dev call did (synchronises): visibilities keyed, distinct kernels built.
int gridhip_aw_last_stats(gridhip_ctx *ctx, int64_t *vis_keyed, int64_t *kernels_built)
{
    if (!ctx) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int32_t h[2] = {0, 0};
    GH_CHECK_HIP(ctx, hipMemcpyAsync(h, ctx->d_scalars + 28, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (kernels_built) *kernels_built = h[0];
    if (vis_keyed) *vis_keyed = h[1];
    return GRIDHIP_OK;
}

// Host pointers: convgrid3 / convgrid4 of src/Gridding.hs:246-396 (same result).
int gridhip_awgrid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q, int64_t S,
                   int64_t A, const double *wkerns, const double *akerns, const double *u, const double *v,
                   int64_t uv_stride, const int64_t *wbin, const int64_t *a1, const int64_t *a2, const double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || W <= 0 || Q <= 0 || S <= 0 || A <= 0 || uv_stride < 1 || !grid || !wkerns ||
        !akerns || (n > 0 && (!u || !v || !wbin || !a1 || !a2 || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)H * Wd, span = n > 0 ? (size_t)(n - 1) * uv_stride + 1 : 1;
    const size_t wel = (size_t)W * Q * Q * S * S, ael = (size_t)A * S * S;
    struct Buf {
        void *p = nullptr;
        ~Buf()
        {
            if (p) (void)hipFree(p);
        }
    } dg, du, dv, dwb, da1, da2, dvis, dwk, dak;
    GH_CHECK_HIP(ctx, hipMalloc(&dg.p, cells * 16));
    GH_CHECK_HIP(ctx, hipMalloc(&du.p, span * 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dv.p, span * 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dwb.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&da1.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&da2.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dvis.p, (size_t)n * 16 + 16));
    GH_CHECK_HIP(ctx, hipMalloc(&dwk.p, wel * 16));
    GH_CHECK_HIP(ctx, hipMalloc(&dak.p, ael * 16));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dg.p, grid, cells * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dwk.p, wkerns, wel * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dak.p, akerns, ael * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n > 0) {
        GH_CHECK_HIP(ctx, hipMemcpyAsync(du.p, u, span * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dv.p, v, span * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dwb.p, wbin, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(da1.p, a1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(da2.p, a2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dvis.p, vis, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    GH_CHECK(gridhip_awgrid_dev(ctx, H, Wd, (double *)dg.p, n, W, Q, S, A, (const double *)dwk.p,
                                (const double *)dak.p, (const double *)du.p, (const double *)dv.p, uv_stride,
                                (const int64_t *)dwb.p, (const int64_t *)da1.p, (const int64_t *)da2.p,
                                (const double *)dvis.p));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(grid, dg.p, cells * 16, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

}  // extern "C"
