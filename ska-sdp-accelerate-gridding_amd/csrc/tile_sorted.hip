// Tap-reusing variant of the tile gridder ("sort" option).
//
// The plain tile kernel streams one [S][S] kernel slice (3.6 KB at 15x15) from L2 for every
// visibility; at 10^8 visibilities that L2->CU stream (360 GB) is what bounds it.  Within one
// work item (w-group x tile, 4000 - 8000 visibilities) only ~1000 distinct slices occur, so this
// variant first orders the work item's records by slice (counting sort: LDS histogram, scan,
// scatter) and each wave then walks a contiguous piece of the sorted list run by run, keeping a
// run's taps in registers and fetching the next runs' taps while the current run is accumulated.
// Tap traffic drops by the mean run length (4 - 8x on the uniform benchmark; far more on real,
// w-coherent data).
//
// Where things live (65 x 89 tiles, 15x15): LDS holds the tile (2 x 65.1 KB, resident for the whole
// work item, flushed once) and the sort's histogram (4.1 KB).  The sorted list itself - 8 bytes per
// record: (slice key | cell offset, orig) - goes to a per-work-group scratch in global memory: it is
// written once and read once, coalesced, by the same CU, and the walkers gather each record's visibility
// value from the caller's array a block of 64 records ahead of its use.  That keeps LDS reads out of the
// accumulate loop altogether: an LDS read returns only after the wave's outstanding ds_add_f64s, so every
// read there would drain the atomic queue.
#include "tile_common.h"

namespace gridhip {

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}

// DEGRID = false: accumulate vis * taps into the tile and flush it onto `grid` (convgrid2).
// DEGRID = true : the tile is loaded from `grid` once and each visibility's sum over taps of
//                 taps * tile is written to vis_out[orig] (degrid2); `vis` is then the output.
// ABL (tuning runs only, option "dbg"): bit 0 = no LDS atomics in the accumulate loop, bit 1 = no flush,
// bit 2 = no tap loads, bit 3 = no visibility gather, bit 4 = per-phase cycle stamps.  Results are wrong
// with any of bits 0..3 set.
//
// Roles.  The work-group's last wave is the SORTER: it pulls the next work item from the queues and
// counting-sorts its records into the other half of the scratch while the remaining waves (WALKERS)
// accumulate the current item; the accumulate loop is bound by the LDS atomic unit, which the sorter
// barely touches, so the sort's latency chains (record load -> LDS histogram slot -> store) disappear
// behind it.  A work-group of one wave does both, one after the other.
constexpr int SORTED_IM_OFF = 65528;  // largest multiple of 8 that a DS instruction's offset field holds

// piece boundaries of the 15 walkers in 1/1024ths of the sorted list.  The SIMD arbiter favours its oldest wave, so with
// equal pieces walkers 0..3 finish at 0.55 of the walk and the last three run on with the LDS unit half idle: the
// pieces are weighted - 1.6 (walkers 0..3), 1.17 (4..7), 0.69 (8..11), 0.40 (12..14) of an equal share where the
// LDS unit binds (rounds of weight / measured finishing time until all walkers were within 4 %: 10.77 -> 10.54 ms at
// cfg3), and the flatter 1.55 / 1.14 / 0.71 / 0.48 where items are sparse and the walkers wait for taps instead
// (Geom.dense = 0: fewer than two visibilities per slice and tile - the steep table costs 0.5 % there - and supports
// below 15, measured 1 - 2 % faster with the flat one).
// [2]: the big tile on sparse items (Geom.dense = 2), re-weighted the same way on the 8192^2 share of configuration 5
__device__ const int cut15[3][16] = {{0, 106, 211, 317, 422, 500, 578, 655, 733, 781, 829, 877, 926, 958, 991, 1024},
                                     {0, 108, 221, 329, 434, 513, 596, 677, 754, 800, 852, 898, 943, 969, 998, 1024},
                                     {0, 100, 202, 303, 401, 480, 562, 641, 718, 769, 824, 875, 925, 957, 992, 1024}};

struct SortedItem {
    int32_t valid, tile, grp, staged;
};

// AW (aw gridders, awgrid.hip): a record's kslice is the index of its de-duplicated (a1, a2, wbin, yf, xf) kernel in a
// table built for this call - arbitrary 31-bit numbers, far more of them than a histogram holds.  The counting sort
// then orders by kslice mod nkeys (equal kernels still end up next to each other unless two of them share a
// residue) and the sorted list carries the kslice itself in place of `orig`, which gridding does not need; a run
// is a stretch of equal kslice.
// BT ("big tile"): the im plane follows the re plane at Geom.imoff bytes, a run-time distance that may exceed what a DS
// instruction's offset field holds - the tile may then use all of the LDS (65 x 110 cells instead of 65 x 89 at
// 15 x 15: a quarter more visibilities per slice and item, what counts where the walk waits for taps) at the price of
// one address add per tap step.
template <int S, bool DEGRID, int ABL = 0, bool AW = false, bool BT = false>
__global__ void __launch_bounds__(1024, 4) tile_grid_sorted_kernel(Geom g, const RecWord *__restrict__ recs,
                                                                const int32_t *__restrict__ bin_start,
                                                                const int32_t *__restrict__ work_start,
                                                                const double2 *__restrict__ gcf,
                                                                double2 *__restrict__ vis,
                                                                double *__restrict__ grid, int nkeys, int batch,
                                                                int32_t *__restrict__ scalars,
                                                                double2 *__restrict__ sorted_vals,
                                                                uint2 *__restrict__ sorted_mo)
{
    extern __shared__ double lds[];
    constexpr int S2 = S * S;
    // Lanes take consecutive taps t = 64 * step + lane of the slice's row-major tap list.
    constexpr int TAIL0 = S2 - ((S2 + 63) / 64 - 1) * 64;  // taps of the list's last step
    // A last step with 33 or 34 taps keeps 32 of them: an LDS atomic with at most two 16-lane groups active costs
    // 6 cycles instead of 7.  The one or two taps left over are added once per block of 64 records, every lane for
    // its own record (gridding only; 15x15: 62 -> 60.25 LDS cycles per visibility).  Large supports whose list ends
    // in one or two taps (31x31: 961 = 15 x 64 + 1) drop that step altogether the same way.
    constexpr int EXTRA = DEGRID ? 0 : (TAIL0 > 32 && TAIL0 <= 34) ? TAIL0 - 32 : (S2 > 256 && TAIL0 <= 2) ? TAIL0 : 0;
    constexpr int S2E = S2 - EXTRA;                 // taps the steps cover
    constexpr int NSTEP_ALL = (S2E + 63) / 64;
    constexpr int TAIL = S2E - (NSTEP_ALL - 1) * 64;
    // Supports above 16 x 16 (more than four steps): a slice's steps are taken in FP PARTS of at most four, so that a
    // part's taps fit the registers a tap set has (3 sets x 4 steps x 4 VGPRs).  The unit the walkers rotate through
    // their tap sets is (run, part): a run's visibilities are accumulated once per part, each time with that part's
    // taps in registers.  One record per visibility, one tile with the full support's halo, no padding - against
    // sub-footprints (api.hip), which give every visibility one record per spatial part.  Parts 0 .. REM-1 have KST
    // steps, the others KST - 1 (all KST when REM = 0); FP = 1 up to 16 x 16.
    // (gridding has the registers for five steps per tap set - 20 VGPRs, 118 in all, no spills - and fewer, longer
    // parts cover the taps' latency better: 17 x 17 as one part of five steps 16.0 -> 13.2 ms against parts of 3 + 2)
    constexpr int MAXST = DEGRID ? 4 : 5;
    constexpr int FP = (NSTEP_ALL + MAXST - 1) / MAXST;
    constexpr int KST = (NSTEP_ALL + FP - 1) / FP;
    constexpr int REM = NSTEP_ALL % FP;
    constexpr int PBASE = NSTEP_ALL / FP;                       // steps of the shorter parts
    constexpr int NSTEP = KST;                                  // registers of one tap set: NSTEP double2
    constexpr int LASTS = (REM == 0 ? KST : KST - 1) - 1;       // the list's last step, as a step of the last part
    // With a 32-tap last step, two visibilities of a run share it: lanes 32..63 hold the same 32 taps again and
    // take the second visibility (one full-width instruction pair, 16 cycles, instead of two half-width ones, 24).
    constexpr bool PAIR = !DEGRID && TAIL == 32;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int plane = g.lrows * g.ldw;
    // LDS: re plane | (gap) | im plane at the fixed byte offset SORTED_IM_OFF | histogram | two item descriptors.
    // The fixed distance lets one address register serve both atomics of a tap: the im one carries the
    // distance in its offset field (16 bits, hence the value), which takes a VALU add per tap step out of
    // the accumulate loop.
    const int IMO = BT ? g.imoff / 8 : SORTED_IM_OFF / 8;  // doubles from a cell's re to its im
    double *lre = lds, *lim = lds + IMO;
    int32_t *hist = reinterpret_cast<int32_t *>(lim + plane);
    const int hist_words = (nkeys + 1 + 3) & ~3;
    SortedItem *desc = reinterpret_cast<SortedItem *>(hist + hist_words);  // [2]
    // this work-group's two halves of the sorted-list scratch
    double2 *svals_wg = sorted_vals + (size_t)blockIdx.x * 2 * batch;
    uint2 *smo_wg = sorted_mo + (size_t)blockIdx.x * 2 * batch;

    // Where a record's visibility value comes from.  w-projection gridding: the walkers gather it themselves from the
    // caller's array, a block of 64 records ahead of its use - the sorted list then carries 8 bytes per record and the
    // values cross the memory system once.  aw gridding: the list's second word is the kernel's index, so the sorter
    // gathers the values and stages them next to the list (16 bytes more per record, written and read back).
    constexpr bool STAGE_VALS = !DEGRID && AW;
    const bool solo = nw == 1;
    const bool is_sorter = wave == nw - 1;
    // the sorter is the youngest wave of its SIMD, i.e. last at the arbiter, and its work is a chain of latencies:
    // it goes first instead (it issues few instructions)
    if (is_sorter && !solo && !(g.dbg & 512)) __builtin_amdgcn_s_setprio(3);
    const bool is_walker = solo || !is_sorter;
    const int nwalk = solo ? 1 : nw - 1;
    const int ncell = g.lrows * g.lcols;

    // shader clock held during this launch (option "clock_khz"): the first work-group stamps the cycle counter and
    // the 100 MHz real-time counter when it starts and when the queues are empty (two scalar stores per launch)
    if (blockIdx.x == 0 && tid == 0) {
        long long *st = reinterpret_cast<long long *>(scalars + 20);
        st[0] = (long long)__builtin_amdgcn_s_memtime();
        st[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    long long prof[7] = {0, 0, 0, 0, 0, 0, 0}, pt = 0;  // ABL & 16: cycles per phase, first lane of the stamping wave
#define GH_STAMP(i_)                        \
    if ((ABL & 16) && lane == 0) {          \
        const long long now_ = clock64();   \
        prof[i_] += now_ - pt;              \
        pt = now_;                          \
    }
    if ((ABL & 16) && lane == 0) pt = clock64();

    // ---- the sorter's job: fetch a work item and leave it sorted in scratch half `slot` -----------------
    // Persistent work-groups: the launch has one work-group per CU slot; each pulls work items of
    // "its" w-group (blockIdx % ngroups: the group whose kernel planes this XCD's L2 holds) from a
    // queue counter, then helps the other groups when its own queue is empty.
    int32_t *queue = scalars + 4;
    int turn = 0;  // sorter state: how many queues this work-group has seen empty
    // Work-groups beyond the persistent ones (option "yield_cus") take a few work items and leave: the CUs they run on
    // come free every few hundred microseconds, which is when a kernel queued on another stream - a collective - can
    // take them; while nothing else is queued the next of these work-groups does.
    int items_left = ((int)blockIdx.x >= g.npersist) ? g.budget : 0x7fffffff;
    auto prepare = [&](int slot) {
        WorkItem w;
        int grp = 0;
        bool have = false;
        if (items_left <= 0) turn = g.ngroups;
        --items_left;
        while (turn < g.ngroups) {
            grp = (blockIdx.x + turn) % g.ngroups;
            int item = 0;
            if (lane == 0) item = atomicAdd(&queue[grp], 1);
            item = __builtin_amdgcn_readfirstlane(item);
            if (find_work_at(g, bin_start, work_start, grp, item, &w)) {
                have = true;
                break;
            }
            ++turn;  // this group's queue is exhausted
        }
        if (!have) {
            if (lane == 0) desc[slot].valid = 0;
            return;
        }
        GH_STAMP(0)  // work fetch
        const int first_plane = (grp * g.W + g.ngroups - 1) / g.ngroups;  // smallest wb with wb*ng/W == grp
        const int first_slice = first_plane * g.Q * g.Q * g.P;
        double2 *svals = svals_wg + (size_t)slot * batch;
        uint2 *smo = smo_wg + (size_t)slot * batch;
        const int b_lo = w.v_lo;
        const int cnt = min(batch, w.v_hi - b_lo);  // a work item never holds more than `batch` records
        // counting sort by kernel slice, one wave: LDS operations of a wave execute in program order; the
        // fences keep the compiler from reordering them across the phases
        for (int i = lane; i <= nkeys; i += 64) hist[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int r0 = lane; r0 < cnt; r0 += 8 * 64) {
            int key[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bool off;  // (a stale record is counted by neither pass)
                key[q] = load_rec(recs, b_lo + min(r0 + q * 64, cnt - 1), g, &off).kslice - first_slice;
                if (AW) key[q] = (int)((unsigned)key[q] % (unsigned)nkeys);
                if (off) key[q] = -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (r0 + q * 64 < cnt && (unsigned)key[q] < (unsigned)nkeys) atomicAdd(&hist[key[q]], 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GH_STAMP(1)  // histogram
        {   // exclusive scan of hist[0..nkeys): each lane owns a contiguous strip
            const int per = (nkeys + 63) / 64;
            const int lo = min(lane * per, nkeys), hi = min(lo + per, nkeys);
            int sum = 0;
            for (int i = lo; i < hi; ++i) sum += hist[i];
            int incl = sum;  // inclusive scan across the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int base = incl - sum;
            for (int i = lo; i < hi; ++i) {
                const int c = hist[i];
                hist[i] = base;
                base += c;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GH_STAMP(2)  // scan
        int bad = 0;
        // eight records per lane and trip: their record loads, then their value gathers, are all in flight
        // together (the gather is a dependent, HBM-random access)
        for (int r0 = lane; r0 < cnt; r0 += 8 * 64) {
            VisRec rec[8];
            double2 val[8];
            bool off[8];  // a field of the record was out of range (stale slot): not staged, counted
#pragma unroll
            for (int q = 0; q < 8; ++q) rec[q] = load_rec(recs, b_lo + min(r0 + q * 64, cnt - 1), g, &off[q]);
            if (STAGE_VALS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) val[q] = (ABL & 8) ? make_double2(1.0, (double)q) : load_nt(vis + rec[q].orig);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (r0 + q * 64 >= cnt) break;
                int key = rec[q].kslice - first_slice;
                if (AW) key = (int)((unsigned)key % (unsigned)nkeys);
                if (off[q] || (unsigned)key >= (unsigned)nkeys) {  // cannot happen unless binning and kernel disagree
                    ++bad;
                    continue;
                }
                // (one returning atomic per record: reserving a run of equal slices in neighbouring lanes together -
                // run_rank, as the pre-pass does - gained 0.7 - 1 % on streams of long tracks and cost the random stream
                // 0.7 %: the extra instructions sit on the sorter's critical path)
                const int pos = atomicAdd(&hist[key], 1);
                if ((unsigned)pos < (unsigned)batch) {
                    // meta = slice key | the footprint origin's cell offset in the tile (< 8192: sorted_plan)
                    smo[pos] = make_uint2(((uint32_t)key << 16) | (uint32_t)((rec[q].lxy >> 16) * g.ldw + (rec[q].lxy & 0xffff)),
                                          (uint32_t)(AW ? rec[q].kslice : rec[q].orig));
                    if (STAGE_VALS) svals[pos] = val[q];
                } else
                    ++bad;
            }
        }
        if (bad) atomicAdd(&scalars[2], bad);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // records that were rejected leave holes at the end of the sorted list: hist[nkeys-1] is now
        // the number actually staged
        if (lane == 0) {
            SortedItem d;
            d.valid = 1;
            d.tile = w.tile;
            d.grp = grp;
            d.staged = hist[nkeys - 1];
            desc[slot] = d;
        }
        GH_STAMP(3)  // scatter + value gather
    };

    int loff[NSTEP];
    const bool tail_ok = lane < TAIL || TAIL == 64;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        int t = s * 64 + lane;
        if (FP == 1 && s == NSTEP - 1 && !tail_ok) t = PAIR ? s * 64 + lane - 32 : 0;  // idle lanes (PAIR: the same taps again)
        loff[s] = (t / S) * g.ldw + (t % S);
    }
    // the tail step's tap index relative to its part's first tap: lanes without a tap repeat a valid one
    const int ttail = tail_ok ? LASTS * 64 + lane : PAIR ? LASTS * 64 + lane - 32 : 0;
    // (FP > 1) this lane's place in a step that starts a row: t = (A + lrow) * S + (B + lcol) for a step whose
    // first tap is A * S + B; lcol + B may carry into the next row
    const int lrow = lane / S, lcol = lane - lrow * S;

    // ---- a walker's job: its piece of the sorted list of scratch half `slot` ------------------------------
    // The piece is taken in blocks of 64 records: two coalesced loads bring a block into registers, one
    // record per lane; runs of equal slice are found with a ballot and every per-record quantity is
    // broadcast with v_readlane, so the accumulate loop issues nothing but arithmetic and ds_add_f64.
    auto walk = [&](int slot, int staged, int first_slice) {
        const double2 *svals = svals_wg + (size_t)slot * batch;
        const uint2 *smo = smo_wg + (size_t)slot * batch;
        // With 15 walkers the pieces are weighted: the SIMD arbiter favours its oldest wave, so with equal pieces
        // walkers 0..3 finish at 0.55 of the walk and the last three run on for the rest with the LDS unit half idle.
        // cut[w] / 1024 = share of the list in front of walker w (option dbg = 256: equal pieces, for comparison).
        int seg_lo, seg_hi;
        if (nwalk == 15 && !DEGRID && !(g.dbg & 256)) {  // (degrid2 measured 0.5 % slower with them)
            seg_lo = (int)(((int64_t)staged * cut15[g.dense][wave]) >> 10);
            seg_hi = (int)(((int64_t)staged * cut15[g.dense][wave + 1]) >> 10);
        } else {
            seg_lo = (int)(((int64_t)staged * wave) / nwalk);
            seg_hi = (int)(((int64_t)staged * (wave + 1)) / nwalk);
        }
        if (seg_lo >= seg_hi) return;
        auto load_list = [&](int b0, uint2 &mo, double2 &v) {
            const int idx = max(min(b0 + lane, seg_hi - 1), seg_lo);  // past the end: the piece's last record
            mo = smo[idx];
            if (STAGE_VALS) v = svals[idx];
        };
        auto load_value = [&](const uint2 &mo, double2 &v) {  // (mo.y = orig, brought below nvis by the sorter)
            if (!DEGRID && !STAGE_VALS) v = (ABL & 8) ? make_double2(1.0, 2.0) : load_nt(vis + mo.y);
        };
        // first step of part `part` in the slice's step list, and how many steps it has
        auto part_step0 = [&](int part) { return FP == 1 ? 0 : part * PBASE + min(part, REM); };
        auto part_nst = [&](int part) { return (FP == 1 || REM == 0 || part < REM) ? KST : KST - 1; };
        auto issue = [&](double2(&k)[NSTEP], int key, int len, int part) {
            key = min(max(key, 0), (AW ? g.nslices : nkeys) - 1);  // never form an address outside the kernel table
            asm volatile("" : "+s"(key));       // (an empty run repeats a key: keep its loads loads, not copies)
            const double2 *kp = gcf + (size_t)(first_slice + key) * S2 + 64 * part_step0(part);
            if (ABL & 4) {
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) k[s] = make_double2((double)key, (double)(lane + s));
                return;
            }
            // the runs of length 0 that pad a block's tail keep the instruction count (and the vmcnt arithmetic)
            // but fetch one element for the whole wave
            const int lofs = len > 0 ? lane : 0, lt = len > 0 ? ttail : 0;
            if (FP == 1) {
#pragma unroll
                for (int s = 0; s < NSTEP - 1; ++s) k[s] = kp[s * 64 + lofs];
                k[NSTEP - 1] = kp[lt];
            } else {
                // always KST loads per unit (the compiler counts them for its vmcnt waits): a part of KST - 1 steps
                // repeats its first element, one address for the whole wave, in the register it does not use
                const bool lastp = part == FP - 1, shortp = part_nst(part) < KST;
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    int idx = s * 64 + lofs;
                    if (s == LASTS && lastp) idx = lt;           // the list's tail step
                    if (s == KST - 1 && shortp) idx = 0;         // no such step in this part
                    k[s] = kp[idx];
                }
            }
        };
        // two blocks of the list and one block of values are in flight ahead of the block being accumulated
        uint2 moN, moNN;
        double2 vN = make_double2(0.0, 0.0), vNN = make_double2(0.0, 0.0);
        load_list(seg_lo, moN, vN);
        load_list(seg_lo + 64, moNN, vNN);
        load_value(moN, vN);
        for (int b0 = seg_lo; b0 < seg_hi; b0 += 64) {
            const uint2 mo = moN;
            const double2 vB = vN;
            moN = moNN;
            vN = vNN;
            load_value(moN, vN);                      // the next block's values (its list entries arrived a block ago) ...
            load_list(b0 + 128, moNN, vNN);           // ... and the list entries of the one after travel meanwhile
            const int bcnt = min(64, seg_hi - b0);
            const uint32_t mykey = AW ? mo.y : mo.x >> 16;
            const uint32_t prevkey = (uint32_t)__shfl_up((int)mykey, 1, 64);
            unsigned long long bits = __ballot(lane < bcnt && (lane == 0 || mykey != prevkey));  // run starts
            double2 kx[EXTRA > 0 ? EXTRA : 1];
            if (EXTRA > 0) {  // this lane's record: the taps its run steps leave out (used after the block's runs)
                const double2 *kp = gcf + (size_t)(first_slice + min((int)mykey, (AW ? g.nslices : nkeys) - 1)) * S2 + (S2 - EXTRA);
#pragma unroll
                for (int e = 0; e < EXTRA; ++e) kx[e] = (ABL & 4) ? make_double2(1.0, 2.0) : kp[e];
            }
            int lastKey = 0;
            int curKey = 0, curStart = 0, curLen = 0, curPart = FP - 1;  // the run whose parts are being handed out
            // next unit of the block: a run - its slice, first lane and length - and which part of the slice's taps
            // (an empty unit once the block is used up)
            auto advance = [&](int &key, int &start, int &len, int &part) {
                if (FP > 1 && curLen > 0 && curPart < FP - 1) {  // the same run, its next part
                    ++curPart;
                    key = curKey;
                    start = curStart;
                    len = curLen;
                    part = curPart;
                    return;
                }
                if (bits == 0) {
                    key = lastKey;
                    start = 0;
                    len = 0;
                    if (FP > 1) part = FP - 1;  // (FP = 1: `part` stays the 0 it was initialised with - written here, the
                    curLen = 0;                 // compiler kept the array of parts in scratch memory: 3.6 GB of stores per launch)
                    return;
                }
                start = (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                len = (bits ? (int)__builtin_ctzll(bits) : bcnt) - start;
                key = lastKey = AW ? __builtin_amdgcn_readlane((int)mo.y, start)
                                   : (int)((uint32_t)__builtin_amdgcn_readlane((int)mo.x, start) >> 16);
                if (FP > 1) {
                    part = 0;
                    curKey = key;
                    curStart = start;
                    curLen = len;
                    curPart = 0;
                }
            };
            auto process = [&](const double2(&k)[NSTEP], int start, int len, int part) {
                // where this unit's taps land relative to a footprint's origin.  FP = 1: the precomputed loff[].
                // FP > 1: step s of the part starts at tap t0 = 64 * (step0 + s) = A * S + B of the list; this lane's
                // tap is (A + lrow) * S + (B + lcol), with a carry into the next row when B + lcol >= S.
                int lo[NSTEP];
                const bool lastp = FP == 1 || part == FP - 1;
                const bool shortp = FP > 1 && part_nst(part) < KST;  // (uniform) the part has KST - 1 steps
                if (FP == 1) {
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) lo[s] = loff[s];
                } else {
                    const int st0 = part_step0(part);
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) {
                        const int tl = 64 * (st0 + s);                 // (scalar) the step's first tap
                        const int A = tl / S, B = tl - A * S;          // (scalar; S is a compile-time constant)
                        int lr = lrow, lc = lcol;
                        if (PAIR && s == LASTS && lastp && lane >= 32) {  // the tail step's second half: the same 32 taps again
                            lr = (lane - 32) / S;
                            lc = (lane - 32) - lr * S;
                        }
                        // (lanes of a non-PAIR tail step beyond its taps point past the footprint: they are switched
                        // off when gridding and read cell lo[0] when degridding)
                        const int carry = (B + lc >= S) ? g.ldw - S : 0;
                        lo[s] = (A + lr) * g.ldw + B + lc + carry;
                    }
                }
                if (!DEGRID) {
                    int i = 0;
                    if (PAIR && lastp) {
                        for (; i + 1 < len; i += 2) {
                            const int j0 = start + i, j1 = j0 + 1;
                            const int lb0 = (int)((uint32_t)__builtin_amdgcn_readlane((int)mo.x, j0) & 0xffff);
                            const int lb1 = (int)((uint32_t)__builtin_amdgcn_readlane((int)mo.x, j1) & 0xffff);
                            const double vx0 = readlane_f64(vB.x, j0), vy0 = readlane_f64(vB.y, j0);
                            const double vx1 = readlane_f64(vB.x, j1), vy1 = readlane_f64(vB.y, j1);
#pragma unroll
                            for (int s = 0; s < LASTS; ++s) {
                                double *cell = lre + (lb0 + lo[s]);
                                if (ABL & 1) continue;
                                __hip_atomic_fetch_add(cell, vx0 * k[s].x - vy0 * k[s].y, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(cell + IMO, vx0 * k[s].y + vy0 * k[s].x,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
#pragma unroll
                            for (int s = 0; s < LASTS; ++s) {
                                double *cell = lre + (lb1 + lo[s]);
                                if (ABL & 1) continue;
                                __hip_atomic_fetch_add(cell, vx1 * k[s].x - vy1 * k[s].y, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(cell + IMO, vx1 * k[s].y + vy1 * k[s].x,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            // the shared last step: each half computes its own visibility's products (EXEC-masked
                            // branches: the empty asm keeps them branches), then all 64 lanes add
                            const double2 kl = k[LASTS];
                            double re, im;
                            int a;
                            if (lane < 32) {
                                re = vx0 * kl.x - vy0 * kl.y;
                                im = vx0 * kl.y + vy0 * kl.x;
                                a = lb0 + lo[LASTS];
                                asm volatile("" : "+v"(a));
                            } else {
                                re = vx1 * kl.x - vy1 * kl.y;
                                im = vx1 * kl.y + vy1 * kl.x;
                                a = lb1 + lo[LASTS];
                                asm volatile("" : "+v"(a));
                            }
                            if (!(ABL & 1)) {
                                __hip_atomic_fetch_add(lre + a, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(lre + a + IMO, im, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    for (; i < len; ++i) {
                        const int j = start + i;
                        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mo.x, j);
                        const int lbase = (int)(m & 0xffff);
                        const double vx = readlane_f64(vB.x, j), vy = readlane_f64(vB.y, j);
#pragma unroll
                        for (int s = 0; s < NSTEP; ++s) {
                            if (FP > 1 && s == KST - 1 && shortp) continue;  // (uniform) this part has no such step
                            const double re = vx * k[s].x - vy * k[s].y;
                            const double im = vx * k[s].y + vy * k[s].x;
                            double *cell = lre + (lbase + lo[s]);
                            if (ABL & 1) {
                                if (re == 1.2345e300 || im == 1.2345e300) *cell = re;  // keeps the arithmetic alive
                                continue;
                            }
                            // the tail step's idle lanes are switched off (EXEC): an LDS atomic costs 8 cycles with
                            // all four 16-lane groups active, 7 with three, 6 with two (tools/micro/lds_atomic.hip)
                            if (TAIL == 64 || s != LASTS || !lastp || tail_ok) {
                                __hip_atomic_fetch_add(cell, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(cell + IMO, im, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                } else {
                    // four visibilities of the run at a time: their partial sums are reduced across the wave
                    // together (wave_sum4_rows), the results land in lanes 15, 31, 47 and 63
                    // (tail step's taps with zeros in the lanes that have none)
                    const double2 kz = tail_ok ? k[LASTS] : make_double2(0.0, 0.0);
                    for (int i = 0; i < len; i += 4) {
                        double sr[4], si[4];
                        int32_t oo[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sr[q] = 0.0;
                            si[q] = 0.0;
                            oo[q] = 0;
                            if (i + q < len) {  // uniform
                                const int j = start + i + q;
                                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mo.x, j);
                                const int lbase = (int)(m & 0xffff);
                                oo[q] = __builtin_amdgcn_readlane((int)mo.y, j);
                                // all of the visibility's tile cells first (2 x NSTEP LDS reads in flight), then the
                                // products: the reads' latency is paid once per visibility, not once per step.  The
                                // lanes without a tap in the tail step read a valid cell and multiply by a zero tap
                                // (kz) instead of being switched off.
                                double gr[NSTEP], gi[NSTEP];
#pragma unroll
                                for (int s = 0; s < NSTEP; ++s) {
                                    const bool nostep = FP > 1 && s == KST - 1 && shortp;  // (uniform)
                                    const bool idle = TAIL != 64 && s == LASTS && lastp && !tail_ok;
                                    const double *cell = lre + (lbase + ((idle || nostep) ? lo[0] : lo[s]));
                                    gr[s] = cell[0];
                                    gi[s] = cell[IMO];
                                }
#pragma unroll
                                for (int s = 0; s < NSTEP; ++s) {
                                    double2 kk = (TAIL != 64 && s == LASTS && lastp) ? kz : k[s];
                                    if (FP > 1 && s == KST - 1 && shortp) kk = make_double2(0.0, 0.0);
                                    // four FMAs (written out: `sr += a*b - c*d` compiles to two multiplies, an
                                    // FMA and an add per component, and this loop is bound by vector-ALU issue)
                                    sr[q] = fma(kk.x, gr[s], sr[q]);
                                    sr[q] = fma(-kk.y, gi[s], sr[q]);
                                    si[q] = fma(kk.x, gi[s], si[q]);
                                    si[q] = fma(kk.y, gr[s], si[q]);
                                }
                            }
                        }
                        const double rr = wave_sum4_rows(sr[0], sr[1], sr[2], sr[3]);
                        const double ri = wave_sum4_rows(si[0], si[1], si[2], si[3]);
                        const int row = lane >> 4;
                        if ((lane & 15) == 15 && i + row < len) {
                            const int32_t o = row == 0 ? oo[0] : row == 1 ? oo[1] : row == 2 ? oo[2] : oo[3];
                            if (g.P == 1 && FP == 1)  // (written once, never read here: past the L2's taps)
                                __builtin_nontemporal_store(dvec2_t{rr, ri}, reinterpret_cast<dvec2_t *>(vis + o));
                            else {  // sub-footprints, parts of the tap list: a visibility's parts are summed (vis_out was cleared)
                                double *dst = reinterpret_cast<double *>(vis + o);
                                unsafeAtomicAdd(dst, rr);
                                unsafeAtomicAdd(dst + 1, ri);
                            }
                        }
                    }
                }
            };

            // NSETS tap sets used in strict rotation: while unit r is accumulated from one set, the taps of units
            // r+1 .. r+NSETS-1 are in flight into the others.  Three sets, two units ahead (a fourth, which the gridding
            // instantiations have the registers for, measured no faster at 4096^2 and 1 % slower at 8192^2: what the
            // walkers wait for there is the L2's bandwidth, not its latency).  The only loop exit sits after a full
            // trip over the sets and every set is "used" after it, so LLVM cannot sink a prefetch past the exit test;
            // units of length 0 pad the tail.
            constexpr int NSETS = 3;
            double2 kS[NSETS][NSTEP];
            int keyS[NSETS], startS[NSETS], lenS[NSETS], partS[NSETS] = {0, 0, 0};
            static_assert(NSETS == 3, "initialiser of partS");
#pragma unroll
            for (int r = 0; r < NSETS - 1; ++r) {
                advance(keyS[r], startS[r], lenS[r], partS[r]);
                issue(kS[r], keyS[r], lenS[r], partS[r]);
            }
            int done = 0;
            for (;;) {
#pragma unroll
                for (int r = 0; r < NSETS; ++r) {
                    const int nx = (r + NSETS - 1) % NSETS;  // (compile time after unrolling)
                    advance(keyS[nx], startS[nx], lenS[nx], partS[nx]);
                    issue(kS[nx], keyS[nx], lenS[nx], partS[nx]);
                    asm volatile("" ::: "memory");  // compiler fence: the prefetch may not sink below this point
                    __builtin_amdgcn_sched_barrier(0);
                    process(kS[r], startS[r], lenS[r], partS[r]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (FP == 1 || partS[r] == FP - 1) done += lenS[r];  // (a run is done with its last part)
                }
                if (done >= bcnt) break;
            }
            // The last prefetches stay "used" on the exit path (in a branch that is never taken: done equals bcnt
            // here), so LLVM cannot sink them below the exit test, and nothing waits for them.
            if (done > bcnt) {
#pragma unroll
                for (int r = 0; r < NSETS - 1; ++r) asm volatile("" ::"v"(kS[r][0].x), "v"(kS[r][NSTEP - 1].y));
            }
            if (EXTRA > 0 && !(ABL & 1) && lane < bcnt) {
                const int lb = (int)(mo.x & 0xffff);
#pragma unroll
                for (int e = 0; e < EXTRA; ++e) {
                    const int t = S2 - EXTRA + e;
                    double *cell = lre + (lb + (t / S) * g.ldw + (t % S));
                    __hip_atomic_fetch_add(cell, vB.x * kx[e].x - vB.y * kx[e].y, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(cell + IMO, vB.x * kx[e].y + vB.y * kx[e].x, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };

    // flush loop: element e = 2 * cell + component; thread t starts at cell t / 2 and moves nthr / 2 cells per trip
    // (nthr is even)
    const int fl_r0 = (tid >> 1) / g.lcols, fl_c0 = (tid >> 1) - fl_r0 * g.lcols;
    const int fl_dr = (nthr >> 1) / g.lcols, fl_dc = (nthr >> 1) - fl_dr * g.lcols;

    // ---- pipeline: item i is accumulated while item i+1 is fetched and sorted ---------------------------
    if (is_sorter) prepare(0);
    if (!DEGRID) {
        for (int i = tid; i < plane; i += nthr) {
            lre[i] = 0.0;
            lim[i] = 0.0;
        }
    }
    __syncthreads();
    for (int cur = 0;; cur ^= 1) {
        const SortedItem d = desc[cur];
        if (!d.valid) break;  // uniform across the work-group
        const int first_plane = (d.grp * g.W + g.ngroups - 1) / g.ngroups;
        const int first_slice = first_plane * g.Q * g.Q * g.P;
        const int tx = d.tile % g.ntx, ty = d.tile / g.ntx;
        const int64_t ox = (int64_t)tx * g.Tx - g.offx, oy = (int64_t)ty * g.Ty - g.offy;
        if (DEGRID) {
            const double2 *gsrc = reinterpret_cast<const double2 *>(grid);
            for (int c = tid; c < ncell; c += nthr) {
                const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
                const int64_t gx = ox + c_, gy = oy + r_;
                double2 v = make_double2(0.0, 0.0);
                if (gx >= 0 && gy >= 0 && gx < g.Wd && gy < g.H) v = gsrc[gy * g.Wd + gx];
                lre[r_ * g.ldw + c_] = v.x;
                lim[r_ * g.ldw + c_] = v.y;
            }
            __syncthreads();
        }
        const long long walk_t0 = (ABL & 16) ? clock64() : 0;
        if (is_sorter && !solo) prepare(cur ^ 1);
        if (is_walker) {
            if ((ABL & 16) && wave == 0 && lane == 0) pt = clock64();
            walk(cur, d.staged, first_slice);
            if ((ABL & 16) && wave == 0) GH_STAMP(4)  // wave 0's own accumulate walk
        }
        if (solo) prepare(cur ^ 1);
        if ((ABL & 16) && lane == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(scalars + 32) + 8 + wave, (unsigned long long)(clock64() - walk_t0));
        __syncthreads();  // the tile is complete, the next item's sorted list and descriptor are in place
        if ((ABL & 16) && wave == 0) GH_STAMP(5)  // waiting for the slowest wave
        if (DEGRID) continue;

        // ---- flush the cells that exist in the grid, and clear the tile for the next item
        // Consecutive lanes take (re, im) of consecutive cells, so one atomic instruction covers a
        // contiguous 512-byte run of the interleaved grid row (memory-side fp64 atomics run at full rate
        // on contiguous runs and at half of it on the stride-16 pattern of one component at a time).
        {   // (row, column) of element e advance by constants from trip to trip: no division in the loop
            const int comp = tid & 1;
            int r_ = fl_r0, c_ = fl_c0;
            for (int e = tid; e < 2 * ncell; e += nthr) {
                const int64_t gx = ox + c_, gy = oy + r_;
                double *cell = (comp ? lim : lre) + r_ * g.ldw + c_;
                c_ += fl_dc;
                r_ += fl_dr;
                if (c_ >= g.lcols) {
                    c_ -= g.lcols;
                    ++r_;
                }
                const double val = *cell;
                if (val == 0.0) continue;
                *cell = 0.0;
                if ((ABL & 2) || gx < 0 || gy < 0 || gx >= g.Wd || gy >= g.H) continue;
                unsafeAtomicAdd(grid + 2 * (gy * g.Wd + gx) + comp, val);
            }
        }
        __syncthreads();
        if ((ABL & 16) && wave == 0) GH_STAMP(6)  // flush
    }
    if (blockIdx.x == 0 && tid == 0) {
        long long *st = reinterpret_cast<long long *>(scalars + 20);
        st[2] = (long long)__builtin_amdgcn_s_memtime();
        st[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if ((ABL & 16) && lane == 0 && (wave == 0 || (is_sorter && !solo))) {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(scalars + 32);
        for (int i = 0; i < 7; ++i)
            if (prof[i]) atomicAdd(&out[i], (unsigned long long)prof[i]);
    }
#undef GH_STAMP
}

// Can the sorted variant run this geometry?  Needs a compile-time support, the per-group slice
// count to fit the 16-bit key and the tile coordinates 8 bits each, and room next to the tile for the
// histogram.  On success returns the LDS bytes and the largest work item (records) it takes.
bool sorted_plan(const gridhip_ctx *ctx, const Geom &g, int block, int *nkeys, int *batch, size_t *lds_bytes)
{
    // square supports with a compile-time instantiation below (the aw form: up to 16)
    if (g.gh != g.gw || g.gh < 5 || g.gh > (g.per_vis ? 16 : 32)) return false;
    const int planes = (g.W + g.ngroups - 1) / g.ngroups + 1;  // groups differ by at most one plane
    // (aw gridders: the sort is by kslice mod 4096)
    const int64_t keys = g.per_vis ? 4096 : (int64_t)planes * g.Q * g.Q * g.P;
    if (keys >= 65536) return false;
    const size_t plane = (size_t)g.lrows * g.ldw * 8;
    if (plane > (size_t)(g.imoff > 0 ? g.imoff : SORTED_IM_OFF)) return false;
    const size_t tile = (size_t)(g.imoff > 0 ? g.imoff : SORTED_IM_OFF) + plane;  // re plane, (gap,) im plane
    const size_t hist = (size_t)((keys + 1 + 3) & ~3) * 4;
    if (tile + hist + 128 > (size_t)ctx->max_lds) return false;
    (void)block;
    int c = ctx->opt.chunk ? (int)ctx->opt.chunk : 16384;  // (dense bins: 16384 measured 1.4 % faster than 8192)
    c = c < 64 ? 64 : c > 16384 ? 16384 : c;
    *nkeys = (int)keys;
    *batch = c;
    *lds_bytes = tile + hist + 128;  // + the item descriptors
    return true;
}

int launch_tile_grid_sorted(gridhip_ctx *ctx, const Geom &g_in, int block, size_t lds_bytes, int nkeys, int batch,
                            int64_t n, const double *gcf, const double *vis, double *grid, bool degrid)
{
    Geom g = g_in;
    Tables t = tables_of(ctx, g);
    const RecWord *recs = (const RecWord *)ctx->recs.ptr;
    if (g.chunk > batch) return fail(ctx, GRIDHIP_EINVAL, "sorted kernel: chunk %d exceeds its work-item capacity %d", g.chunk, batch);
    if (g.per_vis && degrid) return fail(ctx, GRIDHIP_EUNSUPPORTED, "no degrid form of the aw tile kernel");
    if (g.per_vis && g.imoff > 0) return fail(ctx, GRIDHIP_EUNSUPPORTED, "no big-tile form of the aw tile kernel");
    // persistent work-groups: as many as can be resident (LDS-limited), pulling items from per-group queues
    int per_cu = (int)((size_t)ctx->max_lds / lds_bytes);
    per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
    if (per_cu * block > 2048) per_cu = 2048 / block > 0 ? 2048 / block : 1;
    // option "reserve_cus": leave k compute units without a work-group, for a collective queued on another stream
    int cus = ctx->num_cu - (int)ctx->opt.reserve_cus;
    if (cus < g.ngroups) cus = g.ngroups;
    int nblk = cus * per_cu;
    nblk = ctx->opt.reserve_cus ? (nblk / g.ngroups) * g.ngroups : ((nblk + g.ngroups - 1) / g.ngroups) * g.ngroups;
    const int most = work_blocks(g, n);
    if (nblk > most) nblk = most > g.ngroups ? (most / g.ngroups) * g.ngroups : g.ngroups;
    g.npersist = nblk;
    g.budget = 0;
    // option "yield_cus" = k: k CUs' worth of the work-groups are not persistent - instead, up to 2048 further
    // work-groups of 8 work items each are launched, which the dispatcher starts as CUs come free
    if (ctx->opt.yield_cus > 0 && !ctx->opt.reserve_cus) {
        // (rounded up to one CU per shader engine of every XCD - 32 on this part: with fewer the dispatcher's rotation over
        // the engines stops at one without a free CU, no successor is started and the CUs idle like a reservation)
        int k = (((int)ctx->opt.yield_cus + 31) / 32) * 32 * per_cu;
        k = (k / g.ngroups) * g.ngroups;
        if (k > 0 && nblk - k >= g.ngroups) {
            g.npersist = nblk - k;
            g.budget = 8;
            int extra = most - g.npersist;
            extra = extra > 2048 ? 2048 : extra;
            extra = (extra / g.ngroups) * g.ngroups;
            nblk = g.npersist + (extra > k ? extra : k);
        }
    }
    const dim3 gr(nblk), bl(block);
    // two sorted lists (current item, next item) per resident work-group: 8 B (meta, orig) per record, and for the aw
    // gridders, whose sorter stages the values, 16 B more
    const size_t nlist = (size_t)nblk * 2 * batch, nvals = g.per_vis ? nlist : 0;
    GH_CHECK(ws_reserve(ctx, ctx->sorted, nvals * 16 + nlist * 8));
    double2 *svals = (double2 *)ctx->sorted.ptr;  // (not dereferenced when nothing is staged)
    uint2 *smo = (uint2 *)(svals + nvals);
    launch_clear(ctx, t.scalars + 4, 16, (g.dbg & 16) ? t.scalars + 32 : nullptr, (g.dbg & 16) ? 64 : 0);  // the w-groups' queues
#define GH_LAUNCH_(S_, D_, B_)                                                                                   \
    do {                                                                                                         \
        GH_CHECK(raise_lds(ctx, tile_grid_sorted_kernel<S_, D_, 0, false, B_>));                                 \
        hipLaunchKernelGGL((tile_grid_sorted_kernel<S_, D_, 0, false, B_>), gr, bl, lds_bytes, ctx->stream, g, recs, \
                           t.bin_start, t.work_start, (const double2 *)gcf, (double2 *)vis, grid, nkeys, batch, \
                           t.scalars, svals, smo);                                                               \
    } while (0)
#define GH_LAUNCH(S_, D_)                \
    do {                                 \
        if (g.imoff > 0)                 \
            GH_LAUNCH_(S_, D_, true);    \
        else                             \
            GH_LAUNCH_(S_, D_, false);   \
    } while (0)
#define GH_LAUNCH_AW(S_)                                                                                         \
    do {                                                                                                         \
        GH_CHECK(raise_lds(ctx, tile_grid_sorted_kernel<S_, false, 0, true>));                                   \
        hipLaunchKernelGGL((tile_grid_sorted_kernel<S_, false, 0, true>), gr, bl, lds_bytes, ctx->stream, g, recs, \
                           t.bin_start, t.work_start, (const double2 *)gcf, (double2 *)vis, grid, nkeys, batch, \
                           t.scalars, svals, smo);                                                               \
    } while (0)
#define GH_CASE(S_)             \
    case S_:                    \
        if (g.per_vis)          \
            GH_LAUNCH_AW(S_);   \
        else if (degrid)        \
            GH_LAUNCH(S_, true); \
        else                    \
            GH_LAUNCH(S_, false); \
        break;
#define GH_BIG(S_)                                                                  \
    case S_:                                                                        \
        if (g.per_vis) return fail(ctx, GRIDHIP_EUNSUPPORTED, "aw tile kernel: support %d", g.gh); \
        if (degrid)                                                                 \
            GH_LAUNCH(S_, true);                                                    \
        else                                                                        \
            GH_LAUNCH(S_, false);                                                   \
        break;
#define GH_ABL_BT(A_)                                                                                            \
    if (g.gh == 15 && !degrid && g.imoff > 0 && g.dbg == A_) {                                                   \
        GH_CHECK(raise_lds(ctx, tile_grid_sorted_kernel<15, false, A_, false, true>));                           \
        hipLaunchKernelGGL((tile_grid_sorted_kernel<15, false, A_, false, true>), gr, bl, lds_bytes, ctx->stream, g, recs, \
                           t.bin_start, t.work_start, (const double2 *)gcf, (double2 *)vis, grid, nkeys, batch, \
                           t.scalars, svals, smo);                                                               \
        GH_CHECK_HIP(ctx, hipGetLastError());                                                                    \
        return GRIDHIP_OK;                                                                                       \
    }
#define GH_ABL(A_)                                                                                               \
    if (g.gh == 15 && !degrid && g.imoff == 0 && g.dbg == A_) {                                                  \
        GH_CHECK(raise_lds(ctx, tile_grid_sorted_kernel<15, false, A_>));                                        \
        hipLaunchKernelGGL((tile_grid_sorted_kernel<15, false, A_>), gr, bl, lds_bytes, ctx->stream, g, recs,    \
                           t.bin_start, t.work_start, (const double2 *)gcf, (double2 *)vis, grid, nkeys, batch, \
                           t.scalars, svals, smo);                                                               \
        GH_CHECK_HIP(ctx, hipGetLastError());                                                                    \
        return GRIDHIP_OK;                                                                                       \
    }
#ifdef GRIDHIP_TUNING  // ablation / phase-profile instantiations: tuning builds only (make tuning)
    GH_ABL(1) GH_ABL(2) GH_ABL(3) GH_ABL(4) GH_ABL(5) GH_ABL(7) GH_ABL(8) GH_ABL(15) GH_ABL(16) GH_ABL_BT(16)
#endif
#undef GH_ABL
#undef GH_ABL_BT
    switch (g.gh) {
        GH_CASE(5) GH_CASE(6) GH_CASE(7) GH_CASE(8) GH_CASE(9) GH_CASE(10) GH_CASE(11) GH_CASE(12) GH_CASE(13)
        GH_CASE(14) GH_CASE(15) GH_CASE(16)
        GH_BIG(17) GH_BIG(18) GH_BIG(19) GH_BIG(20) GH_BIG(21) GH_BIG(22) GH_BIG(23) GH_BIG(24) GH_BIG(25) GH_BIG(26)
        GH_BIG(27) GH_BIG(28) GH_BIG(29) GH_BIG(30) GH_BIG(31) GH_BIG(32)
        default: return fail(ctx, GRIDHIP_EUNSUPPORTED, "no sorted instantiation for support %d", g.gh);
    }
#undef GH_CASE
#undef GH_BIG
#undef GH_LAUNCH
#undef GH_LAUNCH_
#undef GH_LAUNCH_AW
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
