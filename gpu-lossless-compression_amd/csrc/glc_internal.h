// glc_internal.h -- host-side interfaces between the C-ABI layer (cudpp_api.cpp,
// culzss_api.cpp) and the stage launchers (*.hip).  Plain pointers + sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <functional>

// Diagnostic switches change what a kernel does (extra stores, cycle stamps, debug counters): a library built with one is an
// experiment, never the product.  They compile only with -DGLC_EXPERIMENT_BUILD, and build.py writes such a library (any
// library built with GLC_CXXFLAGS) to GLC_LIB_OUT only -- libglc_amd.so is always the plain build.
#if (defined(GLC_DEBUG_CAND) || defined(GLC_SS_CLOCKS) || defined(GLC_HB_TIMING) || defined(GLC_FS2_CLOCKS) || defined(GLC_EXP_PART) || \
     defined(GLC_EXP_PART2) || defined(GLC_EXP_SORT) || defined(GLC_EXP_MTF)) && !defined(GLC_EXPERIMENT_BUILD)
#error "diagnostic / timing-experiment switches need -DGLC_EXPERIMENT_BUILD (and GLC_LIB_OUT: build.py never writes libglc_amd.so from such a build)"
#endif

namespace glc {

constexpr uint32_t MAX_BLOCK_ELEMS = 1u << 20;   // cudppCompress/BWT limit (cudpp-inpar/README.md:96,99)
constexpr uint32_t HUFF_BLOCK      = 4096;       // HUFF_THREADS_PER_BLOCK*HUFF_WORK_PER_THREAD (cudpp_globals.h:60-61)
constexpr uint32_t HUFF_SYMS       = 257;        // HUFF_NUM_CHARS (cudpp_globals.h:62)
constexpr uint32_t HUFF_MAX_WORDS  = 1536;       // HUFF_CODE_BYTES (cudpp_globals.h:66)
constexpr uint32_t MTF_CHUNK       = 4096;       // bytes of BWT output per wave in the MTF kernels

// fast suffix sorter (bwt_bucket.hip): buckets of FS_AVG suffixes on average, FS_CAP words of slot each
constexpr uint32_t FS_AVG   = 2048;
constexpr uint32_t FS_CAP   = 4096;             // slot size in the word array
constexpr uint32_t FS_FILLMAX = 4032;            // fullest bucket the in-LDS sort takes (a fuller one flags its block)
constexpr uint32_t FS_MAXNB = 512;               // buckets per block at n = 2^20 (256 buckets of 4096 words: k_fs_sort 1.62 vs 1.3 ms)
constexpr uint32_t FS_MAXNB_LOG2 = 9;
// periodic tier (bwt_periodic.hip): blocks that are one periodic stretch with a period of up to PER_PMAX symbols
constexpr uint32_t PER_PMAX = 4096;
constexpr uint32_t PER_TAKE = 256;                   // blocks of one call the periodic tier takes (its scratch, ~140 KB per slot, is sized by this, not by the plan's rows)
constexpr uint32_t PER_Z    = 2;                     // the explicit zone before the break is PER_Z * L symbols wide (bwt_periodic.hip: why 2)
constexpr uint32_t PER_NU   = (2 * PER_Z + 3) * PER_PMAX + 32;   // bytes of the text of representatives: Z L + 2 p + 1 | separator | Z L + t | padding
// L: the multiple of the period p that covers the longer of the period and the tail t
inline __host__ __device__ uint32_t per_span(uint32_t p, uint32_t t) { const uint32_t m = t > p ? t : p; return p * ((m + p - 1) / p); }
inline __host__ __device__ uint32_t per_text_len(uint32_t p, uint32_t t) { return 2 * PER_Z * per_span(p, t) + 2 * p + 2 + t; }
constexpr uint32_t FS_LCP_CAP = 512;             // suffix comparisons and the sample sorter's rounds give up behind this many symbols
#ifndef GLC_SS_TOL_CAP
#define GLC_SS_TOL_CAP 64                        // (128 until round 6.  With chain groups and the capped direct count the doubling rounds are cheap
#endif                                           //  enough to start earlier: a periodic stretch inside Zipf data 6.4 -> 5.5 ms per 32 blocks, bench.py's
                                                 //  partly_deep batch 9.15 -> 8.93 per 64; 48: 5.3 / 9.0, 32: 5.1 / 9.7, 256: 8.1 / 10.2)
constexpr uint32_t SS_TOL_CAP = GLC_SS_TOL_CAP;  // ... and in the sample sorter's tolerant form STOP behind this many: prefix doubling takes over from there (a multiple of 8)
constexpr uint32_t GRP_SAME = 0x80000000u;        // ... this row continues the group of the row before it (k_grp_flags' result; set up front for the members of a run left at the cap)
constexpr uint32_t SA_CAND = 0x40000000u;         // tolerant form, in the rows of s.sa: this row may share SS_TOL_CAP symbols with the row before it (it was still in a
                                                  // run at that depth, or it is a bucket's first row); every other row was told from its neighbours by fewer
#ifndef GLC_SS_LONG
#define GLC_SS_LONG 256
#endif
constexpr uint32_t SS_LONG = GLC_SS_LONG;        // sample sorter: a run of more positions is cut with pivots (k_ss_long), a shorter one counted out (k_ss_windows)
#ifndef GLC_SSL_SMALL
#define GLC_SSL_SMALL 1024
#endif
constexpr uint32_t SSL_PER_BUCKET = FS_FILLMAX / (SS_LONG + 1), SSL_BIG_PER_BUCKET = FS_FILLMAX / (GLC_SSL_SMALL + 1);   // most bins of > SS_LONG / > 1024 members a bucket of <= 4032 can have
static_assert(SSL_PER_BUCKET < 64, "k_ss_cut lists a bucket's long bins with one wave");

// status bits accumulated on the device (PlanBase::d_status)
constexpr uint32_t ST_BLOCK_OVERFLOW = 1u;       // a 4096-symbol block needs > 1536 words
constexpr uint32_t ST_CAPACITY       = 2u;       // compressed stream would not fit its stride
constexpr uint32_t ST_CORRUPT        = 4u;       // decoder: an offset, length or row index of the stream is out of range

// ---------------------------------------------------------------------------
// live per-kernel profile (glcPlanEnableTiming(plan, 3)): hipEvent pairs recorded on the launch stream around
// the launches of the named kernels; durations are folded in when the plan's streams are idle
// ---------------------------------------------------------------------------
enum ProfSlot { PROF_FS_PART = 0, PROF_FS_SORT, PROF_MTF_ENCODE, PROF_HUFF_PACK, PROF_RS_ONESWEEP8, PROF_FS_HIST,
                PROF_MTF_LISTS, PROF_HUFF_BUILD,
                // decoder (decode.hip)
                PROF_DEC_HUFF, PROF_IMTF_POS, PROF_IMTF_REST, PROF_IBWT_LF, PROF_IBWT_WALK, PROF_IBWT_EMIT,
                PROF_NSLOT };
struct KernelProf {
    static constexpr int NPAIR = 4096;
    bool       on = false;
    const char *name[PROF_NSLOT] = {"k_fs_part", "k_fs_sort", "k_mtf_encode", "k_huff_pack", "k_rs_onesweep<8,false>",
                                    "k_fs_hist", "k_mtf_chunk_lists+k_mtf_scan_lists", "k_huff_build",
                                    "k_dec_prepare+k_dec_huff", "k_imtf_pos", "k_imtf_scan+k_imtf_apply",
                                    "k_ibwt_hist+k_rs_scan+k_ibwt_lf", "k_ibwt_walk", "k_ibwt_rank+k_ibwt_emit"};
    double     ms[PROF_NSLOT] = {}, units[PROF_NSLOT] = {};
    long       launches[PROF_NSLOT] = {};
    // event pairs and what they bracket: allocated when profiling is switched on (a plan that never profiles carries
    // none of it)
    hipEvent_t *ev = nullptr;                    // [2 * NPAIR]
    int        *pend_slot = nullptr;             // [NPAIR]
    double     *pend_units = nullptr;            // [NPAIR]
    int        npend = 0;
    long       dropped = 0;                      // launches not bracketed because NPAIR pairs were pending
    long       unread = 0;                       // pairs whose elapsed time could not be read at collect()
    bool enable(bool want)
    {
        if (want && !ev) {
            ev = (hipEvent_t *)calloc(2 * NPAIR, sizeof(hipEvent_t));
            pend_slot = (int *)calloc(NPAIR, sizeof(int));
            pend_units = (double *)calloc(NPAIR, sizeof(double));
            if (!ev || !pend_slot || !pend_units) { on = false; return false; }
        }
        on = want;
        return true;
    }
    int begin(int slot, hipStream_t st)
    {
        if (!on) return -1;
        if (npend >= NPAIR) { dropped++; return -1; }
        const int i = npend;
        for (int k = 0; k < 2; k++)
            if (!ev[2 * i + k] && hipEventCreate(&ev[2 * i + k]) != hipSuccess) return -1;
        pend_slot[i] = slot;
        (void)hipEventRecord(ev[2 * i], st);
        return i;
    }
    void end(int i, double nunits, hipStream_t st)
    {
        if (i < 0) return;
        (void)hipEventRecord(ev[2 * i + 1], st);
        pend_units[i] = nunits;
        npend = i + 1;
    }
    void collect()                               // the caller has synchronised every stream the events were recorded on
    {
        for (int i = 0; i < npend; i++) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) == hipSuccess) {
                ms[pend_slot[i]] += t; units[pend_slot[i]] += pend_units[i]; launches[pend_slot[i]]++;
            } else unread++;
        }
        npend = 0;
    }
    void reset() { for (int k = 0; k < PROF_NSLOT; k++) { ms[k] = 0; units[k] = 0; launches[k] = 0; } npend = 0; dropped = 0; unread = 0; }
    ~KernelProf()
    {
        if (ev) for (int i = 0; i < 2 * NPAIR; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
        free(ev); free(pend_slot); free(pend_units);
    }
};

// C-ABI helper for the stateless device entry points (CULZSS, CUHD-shaped decoder), which have no plan to hang a profile
// on: a process-wide KernelProf with its own slot names.  out3 = {sum of launch ms, launches, units}; returns 0 past the
// last named slot.  Measurement aid: not thread-safe against concurrent launches.
inline int global_prof_get(KernelProf &pr, int nslot, int index, char *name, size_t name_cap, double *out3)
{
    if (index < 0 || index >= nslot || !out3) return 0;
    if (pr.npend) { (void)hipDeviceSynchronize(); pr.collect(); }
    out3[0] = pr.ms[index]; out3[1] = (double)pr.launches[index]; out3[2] = pr.units[index];
    if (name && name_cap) { size_t i = 0; for (; pr.name[index][i] && i + 1 < name_cap; i++) name[i] = pr.name[index][i]; name[i] = 0; }
    return 1;
}

// ---------------------------------------------------------------------------
// suffix array scratch: everything for `rows` blocks of up to nmax elements
// ---------------------------------------------------------------------------
struct SaScratch {
    uint32_t  nmax = 0, rows = 0, max_tiles = 0, rs_tiles = 0;
    uint64_t *keyA = nullptr, *keyB = nullptr;   // [rows][nmax]
    uint32_t *posA = nullptr, *posB = nullptr;   // [rows][nmax] SA slots of the unresolved list
    uint32_t *isa = nullptr;                     // [rows][nmax] rank+1 of each suffix
    uint32_t *sa = nullptr;                      // [rows][nmax]
    uint32_t *tile_hist = nullptr;               // [rows][rs_tiles][512] look-back granules of the radix passes
    uint32_t *digit_base = nullptr;              // [rows][5][512] exclusive digit offsets, one table per pass
    uint32_t *ghist = nullptr;                   // [rows][5][512] digit totals
    uint32_t  epoch = 255;                       // launch tag of the look-back granules (forces a clear first)
    uint64_t *tile_state = nullptr;              // [rows][max_tiles] look-back granules {flag:2, head:21, unres:21, groups:20}
    uint32_t *ticket = nullptr;                  // [rows] tile tickets of the single-pass rank kernel
    uint32_t *hdA = nullptr, *hdB = nullptr;     // [rows][nmax] SA slot of the group head of each unresolved entry
    uint32_t *cntA = nullptr, *cntB = nullptr;   // [rows] unresolved counts
    uint32_t *rl_flag = nullptr, *rl_cnt = nullptr;   // [rows] tile-local refinement: block needs the global sort / its count
    uint32_t *d_max_cnt = nullptr;               // [2] max and sum of the unresolved counts
    uint32_t *h_max_cnt = nullptr;               // pinned [2]
    size_t    bytes = 0;
    bool      force_isa = false;                 // tests: skip text refinement, prefix doubling from round 1
    // fast path (bwt_bucket.hip); its words live in keyA/keyB (one allocation, fs_kstride words per block)
    int       sorter = 0;                        // 0 = bucket sorter, then sample sorter, then general sorter for what each flags;
                                                 // 1 = general sorter only; 2 = general sorter, prefix doubling only;
                                                 // 3 = bucket sorter, then general sorter (no sample sorter);
                                                 // 4 = sample sorter first (a caller that knows its data is text-like)
    size_t    fs_kstride = 0;
    uint32_t *fs_hist = nullptr;                 // [rows][256] symbol counts
    uint2    *fs_tab = nullptr;                  // [rows][256] {C, p} scaled to 2^32
    uint32_t *fs_fill = nullptr, *fs_base = nullptr;   // [rows][FS_MAXNB] bucket fill / rank base
    uint32_t *fs_flag = nullptr;                 // [rows] 1 = a bucket overflowed, 2 = deep (equal codes beyond the depth cap)
    uint32_t *fs_lcnt = nullptr;                 // [rows] n for flagged blocks, 0 otherwise
    uint32_t *fs_redo[2] = {nullptr, nullptr};   // [rows] copies of fs_lcnt, one per call parity (read by the speculative Huffman pass)
    uint32_t *fs_keep[2] = {nullptr, nullptr};   // [rows] 1 = the bucket sorter finished the block (the speculative stages' `only` mask)
    uint32_t *fs_dup = nullptr;                  // [rows] repeated 6-grams among the samples k_fs_hist looks at (text-likeness probe)
    uint32_t *fs_zero = nullptr;                 // [rows] bucket that holds the word of suffix 0 (k_fs_part -> k_fs_sort_bwt: the BWT index is looked for there only)
    uint32_t  parity = 0;                        // set by the caller before sa_build_begin
    uint32_t *fs_nflag = nullptr;                // [8] blocks flagged by the bucket sorter; given up on by the sample sorter; listed for its second attempt; ticket of the finishing kernel; blocks the probe did not call text-like
    uint4    *fs_wl = nullptr;                   // [rows][fs_wl_cap] runs of equal codes: {index << 8 | bwt, first row, first entry, size}
    uint32_t *fs_wlcnt = nullptr;                // [rows] entries in use
    uint32_t  fs_wl_cap = 0;
    uint32_t  last_flagged = 0;                  // blocks of the last sa_build the bucket sorter gave up on
    uint32_t  last_retried = 0;                  // ... the sample sorter took in a second attempt (a bucket past its slot in the first)
    uint32_t  last_general = 0;                  // ... of which the sample sorter gave up on too (general sorter)
    uint32_t  last_resumed = 0;                  // ... of which the doubling rounds RESUMED from the sample sorter's tolerant form
    uint32_t *ss_gtile = nullptr;                // [rows][ceil(nmax / 256)] group heads per tile of rows (k_grp_*)
    uint32_t *ss_cnt2 = nullptr;                 // [rows] n for the blocks the resumed doubling works on, else 0
    // periodic tier: blocks every other tier gave up on that turn out to be ONE periodic stretch (allocated on first use)
    bool      periodic = true;                   // glcPlanSetSorter 7 switches it off
    uint4    *per_info = nullptr;                // [rows] {period, first break, exit smaller?, slot}
    uint32_t *per_list = nullptr, *per_ok = nullptr;   // [rows] taken blocks by slot; whose rows were written
    uint32_t *per_count = nullptr;               // [4] taken, bytes of the longest text of representatives, finished
    uint32_t *per_base = nullptr;                // [min(rows, PER_TAKE)][PER_NU + 1] first row of every representative
    uint8_t  *per_text = nullptr;                // [min(rows, PER_TAKE)][PER_NU] the texts of representatives
    uint32_t  last_periodic = 0;                 // blocks of the last sa_build this tier finished
    uint32_t  resume_min = 2;                    // fewest blocks given up on for depth that are worth the tolerant pass (0: never; sorter modes 5 / 6).  4 until round 6;
                                                 // with chain groups and the tolerant cap at 64: 2 blocks 2.1-2.6 -> 1.6-2.1 ms per call, 3 blocks 2.4-2.9 -> 1.7-2.2; a lone
                                                 // block 1.8-2.3 against 1.6-2.7 (three kinds of four gain, log lines with runs lose: left as it was)
    bool      skip_tier1 = false;                // sorter 4: no bucket-sorter attempt, every block goes to the sample sorter
    // ... and, adaptively, for SMALL calls (sa_skips_tier1): the reference's callers hand over one block per call, and a text
    // block's call spent 0.13 of its 0.69 ms on the bucket sorter's fourteen launches that find the block flagged.  After
    // TEXT_STREAK calls in a row in which the probe called every block text-like, a call of up to TEXT_SKIP_MAX blocks goes
    // straight to the sample sorter; the first block the probe does not call text-like (it is evaluated in skipped calls too)
    // ends the streak.  A wrong guess costs time (that one call's blocks take the sample sorter), never correctness.
    uint32_t  textlike_streak = 0;
    bool      last_skipped = false;              // the plan's last sa_build skipped the bucket sorter's attempt
    // second tier (bwt_bucket.hip, string sample sort): the blocks the bucket sorter flagged
    uint32_t *ss_list = nullptr;                 // [3 rows] their block numbers; behind them the ones that get a second attempt; then the ones for the tolerant form
    uint64_t *ss_split = nullptr;                // [rows][FS_MAXNB] first suffix of every bucket as a word [code : 36 | index : 20 | 0 : 8]
    uint32_t *ss_flag = nullptr;                 // [rows] this tier's give-up flags
    uint32_t *ss_l0 = nullptr;                   // [rows][FS_MAXNB] common prefix of a bucket's two splitters
    uint2    *ss_long = nullptr;                 // [rows * FS_MAXNB * (SSL_PER_BUCKET + SSL_BIG_PER_BUCKET)] long bins of the sample sorter's first cut
    unsigned long long *ss_long_count = nullptr; // their number: small ones in the low half, big ones in the high half
    uint16_t *ss_cell = nullptr;                 // [rows][4098] first splitter of every cell of the code space
    hipEvent_t ev_flag = nullptr;                // marks the readback of fs_nflag (sa_build_begin / sa_build_finish)
    // The sample sorter's SECOND attempt (a block in a few hundred, whose first samples left a bucket past its slot) is a chain
    // of small launches on one block -- 0.54 of a 256-text-block call's 11.3 ms with the chip idle.  A caller that has stages
    // behind the sort (cudpp_api.cpp: MTF + Huffman) sets stage_partial: sa_build_finish calls it ONCE, with a side stream
    // forked off `st` and the mask of the blocks the first attempt finished, before it queues the second attempt, and joins the
    // side stream before it returns; the caller then runs its stages for the blocks of ss_mask[1] only.
    std::function<hipError_t(hipStream_t, const uint32_t *)> stage_partial;
    bool      partial_used = false;              // set by sa_build_finish when it called stage_partial
    uint32_t *ss_mask[2] = {nullptr, nullptr};   // [rows] 1 = flagged block finished by the sample sorter's first attempt / n = still open then
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool      pending = false;
    KernelProf *prof = nullptr;                  // owned by the plan
};

// (a streak of EIGHT: a wrong guess -- an i.i.d. block behind the streak -- costs that call 0.33 ms, a right one saves 0.05-0.09;
//  with two, a stream of three text blocks and one other per cycle lost 12 %; with eight a mixed stream rarely gets there and a
//  text stream is there after eight calls)
constexpr uint32_t TEXT_STREAK = 8, TEXT_SKIP_MAX = 4;
inline bool sa_skips_tier1(const SaScratch &s, uint32_t nblk)
{
    return s.sorter == 4 || (s.sorter == 0 && nblk <= TEXT_SKIP_MAX && s.textlike_streak >= TEXT_STREAK);
}
hipError_t sa_scratch_alloc(SaScratch &s, uint32_t nmax, uint32_t rows);
hipError_t sa_general_reserve(SaScratch &s, bool only_sa);
void       sa_scratch_free(SaScratch &s);

// Suffix arrays of `nblk` blocks of n bytes (block b at text + b*text_stride).
// Result in s.sa[b*nmax ..].  Synchronises the stream once per doubling round.
// If bwt_out != nullptr the BWT bytes (L[i] = SA[i]==0 ? T[n-1] : T[SA[i]-1]) and d_index[b] are
// produced on the way (bwt_compute_final_kernel, compress_kernel.cuh:55-74) -- no separate gather.
hipError_t sa_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                    SaScratch &s, uint8_t *bwt_out = nullptr, size_t bwt_stride = 0, int *d_index = nullptr,
                    int *rounds_out = nullptr);

// two-phase form of sa_build (see bwt_sa.hip): stages that follow the sort can be queued between the two calls
hipError_t sa_build_begin(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                          SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index);
hipError_t sa_build_finish(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                           SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *nflagged);

// the fast path alone (bwt_bucket.hip): enqueues only; flagged blocks are reported in s.fs_lcnt / s.fs_nflag
hipError_t fs_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk, SaScratch &s,
                    uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *sa_out);
uint32_t   fs_bucket_log2(uint32_t n);
// second tier for the nflag blocks listed in s.ss_list: enqueues only; blocks it gives up on keep n in s.fs_lcnt
// (the others get 0) and are counted in s.fs_nflag[1]
hipError_t ss_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nflag, SaScratch &s,
                    uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *sa_out, uint32_t attempt = 0);
// blocks of the first attempt whose only trouble was a bucket past its slot -> listed behind ss_list, count in s.fs_nflag[2]
// (count_only: nothing is listed or cleared -- how many there are decides whether the attempt is worth making)
hipError_t ss_retry_prepare(hipStream_t st, uint32_t nflag, SaScratch &s, uint32_t to = 1, bool count_only = false);
// s.ss_mask[0][b] = 1 for the flagged blocks (redo[b] != 0) the first attempt finished (s.fs_lcnt[b] == 0), s.ss_mask[1] = a
// snapshot of s.fs_lcnt: the blocks still open at that moment
hipError_t ss_split_masks(hipStream_t st, uint32_t nblk, SaScratch &s);

// periodic tier (bwt_periodic.hip); enqueue only.  per_detect lists the taken blocks of s.ss_list[0 .. nlisted) whose ss_flag is
// raised (count -> s.per_count[0], the longest text of representatives -> s.per_count[1]); per_text writes their texts of representatives (nu bytes each, stride PER_NU); per_expand
// turns the suffix arrays of those texts (s.sa, rows 0 .. nper) into the blocks' BWT rows and indices, clears ss_flag /
// fs_lcnt of every block it finishes and counts them in s.per_count[2]
hipError_t per_reserve(SaScratch &s);
hipError_t per_detect(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nlisted, SaScratch &s);
// before the sample sorter's first attempt: listed blocks whose beginning is periodic for 3/8 of the block or more get ss_flag = 3
hipError_t per_probe(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nlisted, SaScratch &s);
hipError_t per_text(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nper, uint32_t nu, SaScratch &s);
hipError_t per_expand(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nper, uint32_t nu, SaScratch &s,
                      uint8_t *bwt_out, size_t bwt_stride, int *d_index);

// copy SA to the cudppSuffixArray layout (out[0]=n, out[1..n]=SA)
hipError_t sa_export(hipStream_t st, const uint32_t *sa, uint32_t n, uint32_t *out);

// ---------------------------------------------------------------------------
// MTF
// ---------------------------------------------------------------------------
struct MtfScratch {
    KernelProf *prof = nullptr;
    uint32_t nmax = 0, rows = 0, max_chunks = 0;
    uint8_t  *lists = nullptr;       // [rows][max_chunks][256] chunk-local recency lists, then start lists
    uint16_t *lens = nullptr;        // [rows][max_chunks]
    size_t    bytes = 0;
};
hipError_t mtf_scratch_alloc(MtfScratch &s, uint32_t nmax, uint32_t rows);
void       mtf_scratch_free(MtfScratch &s);

// out = MTF(in) per block; if sub_hist != nullptr also writes the histogram of
// each 4096-symbol chunk of the output: sub_hist[b][chunk][256].
hipError_t mtf_forward(hipStream_t st, const uint8_t *in, size_t in_stride, uint32_t n, uint32_t nblk,
                       uint8_t *out, size_t out_stride, MtfScratch &s, uint32_t *sub_hist, const uint32_t *only = nullptr,
                       bool skewed = false);
// (`only`, here and in the Huffman stages: if given, blocks whose entry is 0 are skipped -- the second pass over the
//  blocks a later sorter tier rewrote; `skewed`: those blocks are text-like, rank 0 is most of their output -- the encoder's
//  histogram counts it with a ballot per row instead of sixteen adds to one counter)

// ---------------------------------------------------------------------------
// Huffman
// ---------------------------------------------------------------------------
struct HuffScratch {
    KernelProf *prof = nullptr;
    uint32_t nmax = 0, rows = 0, max_sub = 0;
    uint32_t *sub_hist = nullptr;    // [rows][max_sub][256]
    uint32_t *codes = nullptr;       // [rows][257]
    uint32_t *lens = nullptr;        // [rows][257]  (one word each for aligned LDS staging)
    size_t    bytes = 0;
};
hipError_t huff_scratch_alloc(HuffScratch &s, uint32_t nmax, uint32_t rows);
void       huff_scratch_free(HuffScratch &s);

// sub-block histograms of caller-supplied symbols (stand-alone Huffman entry point)
hipError_t huff_histogram(hipStream_t st, const uint8_t *sym, size_t stride, uint32_t n, uint32_t nblk, HuffScratch &s);
// tree + codes + offsets (writes d_hist[b][256], d_offsets[b*offset_stride..], d_size[b]).  redo_flag (optional):
// blocks with a non-zero entry are going to be encoded again (speculative pass over blocks the bucket sorter has
// flagged): their status bits are not raised.
hipError_t huff_build(hipStream_t st, uint32_t n, uint32_t nblk, HuffScratch &s, uint32_t *d_hist,
                      uint32_t *d_offsets, size_t offset_stride, uint32_t *d_size,
                      size_t capacity_words, uint32_t *d_status, const uint32_t *redo_flag = nullptr,
                      const uint32_t *only = nullptr);
// pack (reads mtf bytes, writes the stream)
//   d_block_off (compact layout): block b is written at d_compressed + d_block_off[b]; capacity_words bounds the array
hipError_t huff_pack(hipStream_t st, const uint8_t *mtf, size_t mtf_stride, uint32_t n, uint32_t nblk,
                     HuffScratch &s, const uint32_t *d_offsets, size_t offset_stride,
                     uint32_t *d_compressed, size_t comp_stride_words, const uint32_t *only = nullptr,
                     const unsigned long long *d_block_off = nullptr, size_t capacity_words = 0);
// d_off[b] = *d_start (0 if null) + sizes of the blocks before b; d_off[nblk] = the end; past capacity_words -> ST_CAPACITY
hipError_t huff_block_offsets(hipStream_t st, const uint32_t *d_sizes, uint32_t nblk, unsigned long long *d_off,
                              const unsigned long long *d_start, size_t capacity_words, uint32_t *d_status);

hipError_t compact_streams(hipStream_t st, const uint32_t *d_comp, size_t stride, const uint32_t *d_sizes,
                           uint32_t nblk, uint32_t *d_out, unsigned long long *d_off);

hipError_t expand_streams(hipStream_t st, const uint32_t *d_in, const unsigned long long *d_off, uint32_t nblk,
                          uint32_t *d_comp, size_t stride, uint32_t *d_sizes, uint32_t *d_status);

// ---------------------------------------------------------------------------
// decoder (round-trip parity only; the reference has no GPU decoder)
// ---------------------------------------------------------------------------
struct DecodeScratch {
    KernelProf *prof = nullptr;
    uint32_t nmax = 0, rows = 0, max_tiles = 0, max_split = 0, max_chunks = 0;
    uint8_t  *mtf = nullptr, *bwt = nullptr;     // [rows][nmax]
    uint8_t  *bwt2 = nullptr;                    // second BWT buffer for stage pipelining
    uint8_t  *ilists = nullptr;                  // [rows][max_chunks][256] iMTF chunk permutations / start lists
    uint32_t *lf = nullptr;                      // [rows][nmax+1]  (symbol << 21) | LF(row)
    uint32_t *lut = nullptr;                     // [rows][4096] 12-bit Huffman decode table
    uint32_t *nodes = nullptr;                   // [rows][513]  tree for codes longer than 12 bits
    uint32_t *tile_hist = nullptr;               // [rows][max_tiles][512]
    uint32_t *digit_base = nullptr;              // [rows][512]
    uint32_t *seg = nullptr;                     // [rows][max_seg] len | next << 9
    int      *seg_pos = nullptr;                 // [rows][max_seg] text position of the segment's first symbol
    uint32_t *seg_count = nullptr;               // [rows] segments in use (static splitters + dynamic)
    uint8_t  *slots = nullptr;                   // [rows][max_seg][256] symbols emitted by each segment
    uint32_t  max_seg = 0;
    size_t    bytes = 0;
};
hipError_t tile_hist_scan9(hipStream_t st, uint32_t *tile_hist, uint32_t count, uint32_t *digit_base,
                           uint32_t max_tiles, uint32_t nblk, uint32_t tile_elems);
hipError_t decode_scratch_alloc(DecodeScratch &s, uint32_t nmax, uint32_t rows);
void       decode_scratch_free(DecodeScratch &s);
// d_block_off (compact layout, nblk + 1 entries): block b's words are d_comp[d_block_off[b] .. d_block_off[b + 1])
hipError_t decode_stage_a(hipStream_t st, const uint32_t *d_hist, const uint32_t *d_offsets, size_t offset_stride,
                          const uint32_t *d_comp, size_t comp_stride_words, uint32_t n, uint32_t nblk, DecodeScratch &s,
                          uint8_t *bwt, uint32_t *d_status, const unsigned long long *d_block_off = nullptr);
hipError_t decode_stage_b(hipStream_t st, const int *d_bwt_index, const uint8_t *bwt, uint8_t *d_out, uint32_t n,
                          uint32_t nblk, DecodeScratch &s, uint32_t *d_status);
hipError_t decode_blocks(hipStream_t st, const int *d_bwt_index, const uint32_t *d_hist,
                         const uint32_t *d_offsets, size_t offset_stride, const uint32_t *d_comp,
                         size_t comp_stride_words, uint8_t *d_out, uint32_t n, uint32_t nblk,
                         DecodeScratch &s, MtfScratch &ms, uint32_t *d_status,
                         const unsigned long long *d_block_off = nullptr);

} // namespace glc
