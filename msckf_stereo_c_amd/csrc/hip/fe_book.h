// fe_book.h — the front-end's per-frame bookkeeping as device code: what cg::ImageProcessor does between and after the two
// track calls of a frame (reference msckf_core/src/image_processor.cpp, lines cited per step):
//
//   fe_book1 (after the temporal + stereo track of the previous features)
//     twoPointRansac            :911-1135  (call sites :482-500, commented out in the reference: quirk Q5; runs when
//                                          MSKF_COMPAT_Q5_NO_RANSAC is cleared) inlier markers of the matched cam0 and cam1 pairs
//     trackFeatures tail        :416-513   survivors in track order, lifetime + 1, grid code, per-cell counts, tracking info
//     addNewFeatures head       :622-688   occupancy of the detector cells, detections in cell order, per-grid-cell sieve to
//                                          grid_max by response (stable), the candidate list of the cells with a vacancy and
//                                          every candidate's position in the reference's FULL candidate list (quirk Q4)
//   fe_book2 (after the stereo track of the candidates)
//     addNewFeatures tail       :690-750   matched candidates ranked per cell by response (stable), vacancy fill, ids in
//                                          ascending cell order from the stream's id counter
//     pruneGridFeatures         :758-768   cells over grid_max keep their grid_max longest-lived features (stable)
//     publish / state rotation  :189-200, :1137-1182   the new grid in flatten order: ids, lifetimes, pixels and the
//                                          undistorted points the message carries
//
// Round 2 did all of this on the host (45 us of CPU per stream and frame, two host waits per frame, the points of every
// track call copied to the device and back).  Here one workgroup per VIO stream runs it between the frame's kernels, the
// grid never leaves the device, and the host receives the published grid once per frame.
//
// The SAME source runs on the host in tests/cpp/fe_book_test.cpp: every step is a sequence of phases, a phase is a loop
// over independent items (FB_FOR) that reads only what earlier phases wrote, and phases are separated by FB_SYNC.  On the
// device the items of a phase are spread over the threads of the workgroup and FB_SYNC is the workgroup barrier; on the
// host the items run one after the other.  Work is parallel over the ITEMS (features, detections, candidates), never a
// serial walk of one thread per grid cell over all of them: an item finds its place among the few members of its own
// cell.  Two kinds of cross-item operations: exclusive scans (exact integers: the device takes a wave-shuffle path, the
// host a loop) and counters bumped with FB_INC (an atomic add on the device) that hand out slots of per-cell member lists.
// The ORDER inside such a list depends on the schedule, but nothing that is read from it does: counts, and ranks under
// total orders (ties broken by the item's own index), which is also how the reference's stable sorts are reproduced.
// The host run is therefore a faithful check of the logic.
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../../include/mskf_types.h"

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define FB_DEVICE 1
#else
#define FB_DEVICE 0
#endif
#if defined(__HIPCC__)
#define FB_FN __host__ __device__ inline
#else
#define FB_FN inline
#endif
#if FB_DEVICE
#define FB_FOR(i, n) for (int i = (int)threadIdx.x; i < (n); i += (int)blockDim.x)
#define FB_SYNC() __syncthreads()
#define FB_NTH ((int)blockDim.x)
#define FB_INC(p) atomicAdd((p), 1)
#else
#define FB_FOR(i, n) for (int i = 0; i < (n); ++i)
#define FB_SYNC() ((void)0)
#define FB_NTH 256
#define FB_INC(p) ((*(p))++)
#endif

#define FB_MAXK 16          // grid_min / grid_max above this: the host keeps the books (ImageProcessor falls back)

struct FeGridArr {          // a feature list in HBM (structure of arrays, `cap` entries each)
    unsigned long long *id;
    int *lifetime, *code;
    float *response;
    mskf_point2f *cam0, *cam1, *und0, *und1;
};

struct FeBookState {        // per stream, persistent in HBM
    unsigned long long next_id;         // ImageProcessor::next_feature_id (Q3: starts at 0)
    int n_prev;                         // features of the published grid (= the previous frame's n_curr)
    int n_tracked, n_det, n_cand, n_new, n_curr;
    int before_tracking, after_tracking, after_matching, after_ransac;   // TrackingInfo (:514-530); after_* survive frames without features (:383)
    int overflow;                       // a capacity was hit (cannot happen with the capacities mskf_stream_create derives; checked by the host)
    unsigned long long ransac_draws;    // ImageProcessor::ransac_draws: numbers drawn so far by the counter-based generator of twoPointRansac
};

struct FeBookDev {          // everything the two kernels need for one stream (built by the host per frame)
    // configuration
    int grid_row, grid_col, grid_min, grid_max, n_codes, n_cells;
    int grid_w, grid_h;                 // pixels per grid cell (:250-251)
    int det_rows, det_cols, det_cw, det_ch;
    int thr_score;                      // fast_threshold * 256: detections need a score above it (:132)
    int q4;                             // MSKF_COMPAT_Q4_RESPONSE_INDEX
    int cap, cand_cap, det_cap;         // capacities: grid lists, candidate lists, detection lists
    unsigned int gen;                   // push generation the cell keys must carry
    // 2-point RANSAC between the tracks (:482-500): on when `ransac` is set
    int ransac, ransac_iters;           // iterations = ceil(log(1 - 0.99) / log(1 - 0.7^2)) (:920-921), computed by the host
    double ransac_thr;                  // processor_config.ransac_threshold (inlier_error)
    double ransac_npu[2];               // 2 / (fx + fy) of cam0, cam1 (:917)
    double R_p_c[2][9];                 // rotation previous -> current frame of cam0, cam1 (integrateImuData, :850-889)
    double *rs_pair;                    // scratch in HBM, 4 x cap doubles: per pair its length and the coefficients of (tx, ty, tz) (:949-1012)
    float *rs_pt;                       // scratch, 4 x cap floats: rotated previous point (x, y) and the two norms rescalePoints sums (:888-908)
    double *rs_scalar;                  // scratch, 16 + 4 x 8 doubles: scale, norm_pixel_unit, ..., the models of the hypotheses
    FeBookState *st;
    FeGridArr prev, tracked, curr;
    // results of the first track call, one per previous feature
    const mskf_point2f *t_out0, *t_out1, *t_und0, *t_und1;
    const uint8_t *t_status;
    // detector: per-cell maxima of this frame's cam0 image as keys gen (8) | score (24) | ~order (32)
    const unsigned long long *cell_keys;
    mskf_point2f *det_pt;               // detections in cell order (:657), n_det
    int *det_score;
    // candidates sent to the second track call, n_cand (device-side count: st->n_cand)
    mskf_point2f *cand_pt;
    int *cand_index;                    // position in the reference's full candidate list (Q4, :698)
    int *cand_score;                    // the candidate's own score (what :698 should have read)
    int *cand_off, *cand_cnt;           // per grid cell: its range in the candidate list (0 candidates for a full cell)
    // results of the second track call, one per candidate
    const mskf_point2f *c_out0, *c_out1, *c_und0, *c_und1;
    const uint8_t *c_status;
    int *cell_count;                    // tracked features per grid code, n_codes (written by fe_book1, read by fe_book2)
    // export: what the host receives (one D2H copy per batch)
    int *x_info;                        // 16 ints: n_curr, n_cand, before/after tracking x4, next_id lo/hi, overflow, n_new, n_det, n_tr, ransac_draws lo/hi
    unsigned long long *x_id;
    int *x_lifetime;
    mskf_point2f *x_cam0, *x_cam1, *x_und0, *x_und1;
};

// Workgroup scratch (LDS on the device), carved from one int array by fe_book_scratch_init.
struct FeBookScratch {
    int *a;                 // max(cap, det_cap): flags / scan values; then detection scores
    int *b;                 // max(cap, det_cap): grid codes of the survivors; then of the detections
    int *c;                 // cand_cap: per candidate, the score it is ranked with (-1: no stereo match); then its rank among the cell's new features
    int *d;                 // cap: lifetimes of the survivors
    int *e;                 // cap: RANSAC markers of the cam1 pairs
    int *tl;                // cap: survivors listed by grid cell (order inside a cell unspecified)
    int *dl;                // det_cap: detections listed by grid cell (order inside a cell unspecified)
    int *chunk;             // FB_NTH + 1: partial sums of the scans, [FB_NTH] = total
    int *cell[8];           // n_codes + 1 each
    int *rs;                // 16: support counts of the RANSAC hypotheses
    unsigned char *occ;     // det_rows * det_cols
};

FB_FN size_t fe_book_scratch_ints(int cap, int cand_cap, int det_cap, int n_codes, int det_cells) {
    const int m = cap > det_cap ? cap : det_cap;
    return (size_t)2 * m + cand_cap + 3 * (size_t)cap + det_cap + (FB_NTH + 1) + (size_t)8 * (n_codes + 1) + 16 + (size_t)(det_cells + 3) / 4 + 8;
}
FB_FN void fe_book_scratch_init(FeBookScratch &L, int *mem, int cap, int cand_cap, int det_cap, int n_codes, int det_cells) {
    const int m = cap > det_cap ? cap : det_cap;
    L.a = mem; mem += m;
    L.b = mem; mem += m;
    L.c = mem; mem += cand_cap;
    L.d = mem; mem += cap;
    L.e = mem; mem += cap;
    L.tl = mem; mem += cap;
    L.dl = mem; mem += det_cap;
    L.chunk = mem; mem += FB_NTH + 1;
    for (int q = 0; q < 8; ++q) { L.cell[q] = mem; mem += n_codes + 1; }
    L.rs = mem; mem += 16;
    L.occ = (unsigned char *)mem;
    (void)det_cells;
}

// image_processor.cpp:452-454 (Q7: the column may equal grid_col): float division, truncation
FB_FN int fb_grid_code(const FeBookDev &B, float x, float y) {
    return (int)(y / (float)B.grid_h) * B.grid_col + (int)(x / (float)B.grid_w);
}

// Exclusive prefix sum of v[0, n) in place, total in chunk[FB_NTH] (workgroup-shared); exact integers, so the two
// implementations give the same values: on the device blockDim elements per pass, inclusive scan inside a wavefront with
// shuffles, the waves' totals through `chunk`; on the host a loop.  Ends with a barrier.
FB_FN void fb_exclusive_scan(int *v, int n, int *chunk) {
#if FB_DEVICE
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = (int)blockDim.x, nw = nth >> 6;
    __syncthreads();                 // (everyone has read the previous scan's total before this one replaces it)
    int carry = 0;
    for (int base = 0; base < n; base += nth) {
        const int i = base + tid;
        const int own = i < n ? v[i] : 0;
        int x = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) chunk[wave] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < nw; ++w) { const int t = chunk[w]; if (w < wave) woff += t; tot += t; }
        if (i < n) v[i] = carry + woff + x - own;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) chunk[FB_NTH] = carry;
    __syncthreads();
#else
    int run = 0;
    for (int i = 0; i < n; ++i) { const int t = v[i]; v[i] = run; run += t; }
    chunk[FB_NTH] = run;
#endif
}


// ------------------------------------------------------------------------------------------ 2-point RANSAC (:911-1135)
// cg::uniform_integer lives in the absent vikit_cg: the draws come from a counter-based generator (splitmix64 of the
// stream's draw counter, shared with the host mirror and the oracle), so the j-th number ever drawn is a function of j
// alone and the hypotheses of a call can be formed side by side.
FB_FN int fb_uniform_int(unsigned long long draw, int lo, int hi) {
    unsigned long long z = 0x5EED5EED5EED5EEDULL + 0x9E3779B97F4A7C15ULL * draw;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return lo + (int)(z % (unsigned long long)(hi - lo + 1));
}
// float square root / quotient through double: correctly rounded (53 >= 2 * 24 + 2 bits), whatever the float forms
// of the target compile to
FB_FN float fb_sqrtf(float v) { return (float)sqrt((double)v); }
FB_FN float fb_divf(float a, float b) { return (float)((double)a / (double)b); }

// Inlier markers of the `n` matched pairs of camera `cam`: pair k is previous feature list[k], its points are the
// undistorted previous point of the published grid and the undistorted tracked point of this frame.  marker[k] = 1 / 0.
// The arithmetic is the host mirror's (csrc/host/image_processor.cpp two_point_ransac: single precision where the reference
// holds cg::Point2f, double elsewhere); what is a sum over the pairs in INDEX ORDER there - rescalePoints' float sum and the
// mean length - is summed by one item in that order here, everything else is independent per pair or per hypothesis.
// Every item of the workgroup takes the same path (the decisions are read from shared scratch after a barrier).
FB_FN void fb_two_point_ransac(const FeBookDev &B, FeBookScratch &L, int cam, int n, const int *list, int *marker) {
    if (n == 0) return;
    const mskf_point2f *prev_und = cam == 0 ? B.prev.und0 : B.prev.und1;
    const mskf_point2f *curr_und = cam == 0 ? B.t_und0 : B.t_und1;
    const double *R = B.R_p_c[cam];
    const int cap = B.cap;
    double *len = B.rs_pair, *ctx = len + cap, *cty = ctx + cap, *ctz = cty + cap;
    float *px = B.rs_pt, *py = px + cap, *n1 = py + cap, *n2 = n1 + cap;
    double *sc = B.rs_scalar, *model = sc + 16;
    enum { SC_SCALE = 0, SC_NPU, SC_MEAN, SC_CAND, SC_BEST };
    // previous points compensated with the relative rotation (:936-944, no perspective division) and the norms of :893-897
    FB_FOR(k, n) {
        const int i = list[k];
        const mskf_point2f p = prev_und[i], c = curr_und[i];
        const double X = R[0] * (double)p.x + R[1] * (double)p.y + R[2] * 1.0;
        const double Y = R[3] * (double)p.x + R[4] * (double)p.y + R[5] * 1.0;
        const float fx = (float)X, fy = (float)Y;
        px[k] = fx; py[k] = fy;
        n1[k] = fb_sqrtf(fx * fx + fy * fy);
        n2[k] = fb_sqrtf(c.x * c.x + c.y * c.y);
    }
    FB_SYNC();
    FB_FOR(one, 1) {
        float sum = 0.0f;
        for (int k = 0; k < n; ++k) { sum += n1[k]; sum += n2[k]; }
        const float scale = fb_divf((float)(n + n), sum) * fb_sqrtf(2.0f);
        sc[SC_SCALE] = (double)scale;
        sc[SC_NPU] = B.ransac_npu[cam] * (double)scale;
    }
    FB_SYNC();
    const float scale = (float)sc[SC_SCALE];
    const double npu = sc[SC_NPU];
    FB_FOR(k, n) {
        const mskf_point2f c0 = curr_und[list[k]];
        const float ax = px[k] * scale, ay = py[k] * scale, bx = c0.x * scale, by = c0.y * scale;
        const float dx = ax - bx, dy = ay - by;
        const double l = sqrt((double)(dx * dx + dy * dy));
        len[k] = l;
        ctx[k] = (double)dy; cty[k] = (double)(-dx); ctz[k] = (double)(ax * by - ay * bx);
        marker[k] = l > 50.0 * npu ? 0 : 1;                    // :961-967
    }
    FB_SYNC();
    FB_FOR(one, 1) {
        double length_sum = 0.0;
        int candidates = 0;
        for (int k = 0; k < n; ++k) if (marker[k]) { length_sum += len[k]; ++candidates; }
        sc[SC_MEAN] = length_sum / candidates;
        sc[SC_CAND] = (double)candidates;
    }
    FB_SYNC();
    const int candidates = (int)sc[SC_CAND];
    if (candidates < 3) {                                       // :974-977
        FB_FOR(k, n) marker[k] = 0;
        FB_SYNC();
        return;
    }
    const double tol = B.ransac_thr * npu;
    if (sc[SC_MEAN] < npu) {                                    // degenerate (pure rotation) case, :985-1001
        FB_FOR(k, n) if (marker[k] && len[k] > tol) marker[k] = 0;
        FB_SYNC();
        return;
    }
    // the pool of pairs the hypotheses draw from: the marked pairs in index order (L.a: exclusive scan of the markers)
    FB_FOR(k, n) L.a[k] = marker[k];
    FB_SYNC();
    fb_exclusive_scan(L.a, n, L.chunk);
    const int m = L.chunk[FB_NTH];
    FB_SYNC();
    FB_FOR(k, n) if (marker[k]) L.b[L.a[k]] = k;               // pool[j] = j-th marked pair
    const unsigned long long draw0 = B.st->ransac_draws;
    const int iters = B.ransac_iters < 8 ? B.ransac_iters : 8;
    FB_FOR(h, 16) L.rs[h] = 0;
    FB_SYNC();
    // the model of every hypothesis: two distinct pairs (:1025-1033), the coefficient column with the smallest L1 norm is
    // fixed to 1 and the 2 x 2 system gives the other two (:1036-1065)
    FB_FOR(h, iters) {
        const int first = fb_uniform_int(draw0 + 2ULL * (unsigned)h + 1ULL, 0, m - 1);
        const int step = fb_uniform_int(draw0 + 2ULL * (unsigned)h + 2ULL, 1, m - 1);
        const int second = first + step < m ? first + step : first + step - m;
        const int i1 = L.b[first], i2 = L.b[second];
        // (selects instead of arrays indexed by `fixed`: a dynamically indexed array lives in scratch memory on the device)
        const double x1 = ctx[i1], y1 = cty[i1], z1 = ctz[i1], x2 = ctx[i2], y2 = cty[i2], z2 = ctz[i2];
        const double l1x = fabs(x1) + fabs(x2), l1y = fabs(y1) + fabs(y2), l1z = fabs(z1) + fabs(z2);
        int fixed = 0;
        double l1min = l1x;
        if (l1y < l1min) { fixed = 1; l1min = l1y; }
        if (l1z < l1min) { fixed = 2; l1min = l1z; }
        // the two other columns, ascending: (y, z), (x, z) or (x, y)
        const double a0 = fixed == 0 ? y1 : x1, a1 = fixed == 0 ? y2 : x2;
        const double b0 = fixed == 2 ? y1 : z1, b1 = fixed == 2 ? y2 : z2;
        const double f1 = fixed == 0 ? x1 : (fixed == 1 ? y1 : z1), f2 = fixed == 0 ? x2 : (fixed == 1 ? y2 : z2);
        const double r0 = -f1, r1 = -f2;
        const double det = a0 * b1 - b0 * a1;
        const double v00 = b1 / det, v01 = -b0 / det, v10 = -a1 / det, v11 = a0 / det;
        const double ta = v00 * r0 + v01 * r1, tb = v10 * r0 + v11 * r1;
        double t[3];
        t[0] = fixed == 0 ? 1.0 : ta;
        t[1] = fixed == 1 ? 1.0 : (fixed == 0 ? ta : tb);
        t[2] = fixed == 2 ? 1.0 : tb;
        model[4 * h + 0] = t[0]; model[4 * h + 1] = t[1]; model[4 * h + 2] = t[2];
    }
    FB_SYNC();
    // support of every hypothesis (:1067-1076): bit h of L.a[k] = pair k agrees with hypothesis h
    FB_FOR(k, n) {
        int bits = 0;
        if (marker[k])
            for (int h = 0; h < iters; ++h) {
                const double err = (ctx[k] * model[4 * h + 0] + cty[k] * model[4 * h + 1]) + ctz[k] * model[4 * h + 2];
                if (fabs(err) < tol) { bits |= 1 << h; FB_INC(&L.rs[h]); }
            }
        L.a[k] = bits;
    }
    FB_SYNC();
    // the first hypothesis with the largest support wins, hypotheses below 0.2 n are skipped (:1078-1079, :1123-1126; the
    // refit of :1082-1121 only feeds an error nothing reads)
    FB_FOR(one, 1) {
        int best = -1, best_size = 0;
        for (int h = 0; h < iters; ++h) {
            const int sz = L.rs[h];
            if ((double)sz < 0.2 * (double)n) continue;
            if (sz > best_size) { best = h; best_size = sz; }
        }
        sc[SC_BEST] = (double)best;
        B.st->ransac_draws = draw0 + 2ULL * (unsigned)iters;
    }
    FB_SYNC();
    const int best = (int)sc[SC_BEST];
    FB_FOR(k, n) marker[k] = best >= 0 ? ((L.a[k] >> best) & 1) : 0;
    FB_SYNC();
}

// ------------------------------------------------------------------------------------------ after the first track call
FB_FN void fe_book1(const FeBookDev &B, FeBookScratch &L) {
    FeBookState &st = *B.st;
    const int n = st.n_prev;
    const int det_cells = B.det_rows * B.det_cols;
    const int *tot = L.chunk + FB_NTH;
    int *cnt_t = L.cell[0], *det_cnt = L.cell[1], *kept = L.cell[2], *flat = L.cell[3], *c_cnt = L.cell[4], *c_off = L.cell[5], *d_off = L.cell[6], *fill = L.cell[7];
    // ---- trackFeatures tail (:416-513).  status bit 0: temporal track inside the image, bit 1: stereo match accepted (only
    //      ever set together with bit 0).  Survivors keep their order (:440, :465-480 compaction), lifetime + 1 (:508).
    FB_FOR(i, n) { const int s = B.t_status[i]; L.a[i] = (s & 3) == 3 ? 1 : 0; L.b[i] = (s & 1) ? 1 : 0; }
    FB_FOR(c, B.n_codes + 1) { cnt_t[c] = 0; det_cnt[c] = 0; fill[c] = 0; }
    FB_FOR(k, det_cells) L.occ[k] = 0;
    FB_SYNC();
    fb_exclusive_scan(L.b, n, L.chunk);                 // tracking info: features with bit 0
    const int n_bit0 = *tot;
    fb_exclusive_scan(L.a, n, L.chunk);
    const int n_match = *tot;
    FB_SYNC();
    if (B.ransac) {
        // :482-500 (Q5 cleared): the matched pairs of cam0 and of cam1 each go through twoPointRansac, a feature survives as
        // an inlier of both.  L.tl = the matched features in order, L.d / L.e = the two marker lists (tl and d are only used
        // by fe_book2)
        int *list = L.tl, *in0 = L.d, *in1 = L.e;
        FB_FOR(i, n) if ((B.t_status[i] & 3) == 3) list[L.a[i]] = i;
        FB_SYNC();
        fb_two_point_ransac(B, L, 0, n_match, list, in0);
        fb_two_point_ransac(B, L, 1, n_match, list, in1);
        FB_FOR(i, n) L.a[i] = 0;
        FB_SYNC();
        FB_FOR(k, n_match) L.a[list[k]] = (in0[k] && in1[k]) ? 1 : 0;
        FB_SYNC();
        FB_FOR(i, n) L.b[i] = L.a[i];                   // (b held the bit-0 scan: n_bit0 is taken)
        FB_SYNC();
        fb_exclusive_scan(L.a, n, L.chunk);
    }
    const int n_tr = *tot;
    FB_SYNC();
    FB_FOR(i, n) {
        if ((B.t_status[i] & 3) != 3) continue;
        if (B.ransac && !L.b[i]) continue;
        const int k = L.a[i];
        const mskf_point2f p = B.t_out0[i];
        int code = fb_grid_code(B, p.x, p.y);
        if (code < 0) code = 0;
        if (code >= B.n_codes) code = B.n_codes - 1;      // (cannot happen for a point inside the image: n_codes covers every code)
        B.tracked.id[k] = B.prev.id[i];
        B.tracked.lifetime[k] = B.prev.lifetime[i] + 1;
        B.tracked.code[k] = code;
        B.tracked.response[k] = 0.f;
        B.tracked.cam0[k] = p; B.tracked.cam1[k] = B.t_out1[i];
        B.tracked.und0[k] = B.t_und0[i]; B.tracked.und1[k] = B.t_und1[i];
        FB_INC(&cnt_t[code]);                               // survivors per grid code
        // CornerDetector::set_grid_position of the truncated pixel (:632-649)
        const int xi = (int)p.x, yi = (int)p.y;
        int r = (int)((float)yi / (float)B.det_ch), c = (int)((float)xi / (float)B.det_cw);
        r = r < 0 ? 0 : (r >= B.det_rows ? B.det_rows - 1 : r);
        c = c < 0 ? 0 : (c >= B.det_cols ? B.det_cols - 1 : c);
        L.occ[r * B.det_cols + c] = 1;                      // (several items may store the same 1)
    }
    FB_SYNC();
    FB_FOR(c, B.n_codes) B.cell_count[c] = cnt_t[c];
    FB_FOR(one, 1) {
        st.n_tracked = n_tr;
        st.before_tracking = n;
        if (n > 0) { st.after_tracking = n_bit0; st.after_matching = n_match; st.after_ransac = n_tr; }   // (:383: nothing is touched without features)
    }
    // ---- detections (:657): cells in order whose maximum beats the threshold and that hold no live feature
    FB_FOR(k, det_cells) {
        const unsigned long long key = B.cell_keys[k];
        const int score = (unsigned int)(key >> 56) == B.gen ? (int)((key >> 32) & 0xFFFFFFULL) : 0;
        L.a[k] = (score > B.thr_score && !L.occ[k]) ? 1 : 0;
        L.dl[k] = L.a[k];                                   // (dl is free until the per-cell lists are built)
    }
    FB_SYNC();
    fb_exclusive_scan(L.a, det_cells, L.chunk);
    const int n_det = *tot;
    FB_SYNC();
    FB_FOR(k, det_cells) {
        if (!L.dl[k]) continue;
        const unsigned long long key = B.cell_keys[k];
        const unsigned int order = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFULL);
        const int cy = k / B.det_cols, cx = k - cy * B.det_cols;
        const int q = L.a[k];
        mskf_point2f p;
        p.y = (float)(cy * B.det_ch + (int)(order / (unsigned)B.det_cw));
        p.x = (float)(cx * B.det_cw + (int)(order % (unsigned)B.det_cw));
        const int score = (int)((key >> 32) & 0xFFFFFFULL);
        B.det_pt[q] = p;
        B.det_score[q] = score;
        // grid cell of the detection (:661-663; outside the nominal cells: dropped, Q7)
        const int code = fb_grid_code(B, p.x, p.y);
        L.b[q] = code;
        if (code >= 0 && code < B.n_cells) FB_INC(&det_cnt[code]);
    }
    FB_SYNC();
    FB_FOR(q, n_det) L.a[q] = B.det_score[q];               // (a held the scan: every reader of it is past the barrier)
    // ---- sieve (:661-677): every grid cell keeps its grid_max best detections by response, stable (equal responses keep
    //      their detection order); a cell with fewer is not sorted at all (:664 sorts only when it has to cut).  Only the
    //      cells with a vacancy send theirs on (a full cell's candidates cannot influence any output), but every cell's kept
    //      count moves the position in the reference's full candidate list (Q4).
    FB_FOR(c, B.n_cells) {
        const int k = det_cnt[c] < B.grid_max ? det_cnt[c] : B.grid_max;
        kept[c] = k; flat[c] = k;
        const int cc = cnt_t[c] < B.grid_min ? k : 0;
        c_cnt[c] = cc; c_off[c] = cc;
        d_off[c] = det_cnt[c];
    }
    FB_SYNC();
    fb_exclusive_scan(flat, B.n_cells, L.chunk);             // position of the cell's first candidate in the full list
    fb_exclusive_scan(d_off, B.n_cells, L.chunk);            // the cell's range in the per-cell detection lists
    fb_exclusive_scan(c_off, B.n_cells, L.chunk);            // offset of the cell in the list that is sent on
    const int n_cand = *tot;
    FB_SYNC();
    FB_FOR(c, B.n_cells) { B.cand_cnt[c] = c_cnt[c]; B.cand_off[c] = c_off[c]; }
    FB_FOR(q, n_det) {
        const int code = L.b[q];
        if (code < 0 || code >= B.n_cells || c_cnt[code] <= 0) continue;
        L.dl[d_off[code] + FB_INC(&fill[code])] = q;        // member list of the cell (order unspecified)
    }
    FB_SYNC();
    // every detection of a cell with a vacancy finds its own place: its rank among the cell's detections, by (response
    // descending, detection order) when the cell has to cut, by detection order when it does not
    FB_FOR(q, n_det) {
        const int code = L.b[q];
        if (code < 0 || code >= B.n_cells || c_cnt[code] <= 0) continue;
        const int o = d_off[code], m = det_cnt[code];
        const bool cut = m > B.grid_max;
        const int s = L.a[q];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const int q2 = L.dl[o + j];
            const int s2 = L.a[q2];
            rank += cut ? ((s2 > s || (s2 == s && q2 < q)) ? 1 : 0) : (q2 < q ? 1 : 0);
        }
        if (rank >= kept[code]) continue;
        const int at = c_off[code] + rank;
        if (at >= B.cand_cap) { st.overflow = 1; continue; }
        B.cand_pt[at] = B.det_pt[q];
        B.cand_score[at] = s;
        B.cand_index[at] = flat[code] + rank;
    }
    FB_FOR(one, 1) { st.n_det = n_det; st.n_cand = n_cand < B.cand_cap ? n_cand : B.cand_cap; }
    FB_SYNC();
}

// ------------------------------------------------------------------------------------------ after the second track call
FB_FN void fe_book2(const FeBookDev &B, FeBookScratch &L) {
    FeBookState &st = *B.st;
    const int n_tr = st.n_tracked, n_cand = st.n_cand, n_det = st.n_det;
    const int *tot = L.chunk + FB_NTH;
    int *cnt_t = L.cell[0], *t_off = L.cell[1], *fill = L.cell[2], *new_cnt = L.cell[3], *new_off = L.cell[4], *out_off = L.cell[5];
    // codes and lifetimes of the survivors, their per-cell counts, and per candidate the score it is ranked with: under Q4
    // the detection-order score at the candidate's position in the full candidate list (:698), else its own
    FB_FOR(k, n_tr) { L.b[k] = B.tracked.code[k]; L.d[k] = B.tracked.lifetime[k]; }
    FB_FOR(c, B.n_codes + 1) { const int v = c < B.n_codes ? B.cell_count[c] : 0; cnt_t[c] = v; t_off[c] = v; fill[c] = 0; }
    FB_FOR(i, n_cand) {
        int sc = -1;
        if (B.c_status[i] & 2) {
            sc = B.cand_score[i];
            if (B.q4) { const int di = B.cand_index[i]; sc = di < n_det ? B.det_score[di] : 0; }
        }
        L.c[i] = sc;
    }
    FB_SYNC();
    fb_exclusive_scan(t_off, B.n_codes, L.chunk);
    FB_SYNC();
    FB_FOR(k, n_tr) { const int c = L.b[k]; L.tl[t_off[c] + FB_INC(&fill[c])] = k; }     // survivors of every cell (order unspecified)
    // ---- addNewFeatures tail (:700-750): per cell, the matched candidates ranked by response (stable) fill the vacancy.
    //      A candidate's rank among its cell's matched candidates, by (response descending, candidate order): a[i], or -1
    FB_FOR(i, n_cand) {
        L.a[i] = -1;
        const int sc = L.c[i];
        if (sc < 0) continue;
        const mskf_point2f p = B.cand_pt[i];
        const int c = fb_grid_code(B, p.x, p.y);
        if (c < 0 || c >= B.n_cells) continue;
        const float r = (float)((double)sc / 256.0);
        const int o = B.cand_off[c], e = o + B.cand_cnt[c];
        int rank = 0;
        for (int j = o; j < e && j < n_cand; ++j) {
            const int s2 = L.c[j];
            if (s2 < 0) continue;
            const float r2 = (float)((double)s2 / 256.0);
            rank += (r2 > r || (r2 == r && j < i)) ? 1 : 0;
        }
        if (rank < B.grid_min - cnt_t[c]) L.a[i] = rank;
    }
    FB_FOR(c, B.n_codes) {
        int m = 0;
        if (c < B.n_cells) {
            const int o = B.cand_off[c], e = o + B.cand_cnt[c];
            for (int j = o; j < e && j < n_cand; ++j) m += L.c[j] >= 0 ? 1 : 0;
            const int vac = B.grid_min - cnt_t[c];
            m = m < vac ? m : (vac > 0 ? vac : 0);
        }
        new_cnt[c] = m; new_off[c] = m;
        const int total = cnt_t[c] + m;
        out_off[c] = total < B.grid_max ? total : B.grid_max;
    }
    FB_SYNC();
    fb_exclusive_scan(new_off, B.n_codes, L.chunk);          // rank of the cell's first new feature: ids in ascending cell order (:745)
    const int n_new = *tot;
    fb_exclusive_scan(out_off, B.n_codes, L.chunk);          // where the cell starts in the published grid
    const int n_curr = *tot;
    FB_SYNC();
    const unsigned long long id0 = st.next_id;
    // ---- this frame's grid (:498-513, :735-750) and pruneGridFeatures (:758-768): per cell the survivors in track order,
    //      then the new features in rank order; a cell over grid_max keeps its grid_max longest-lived, stable.  A survivor's
    //      slot is its rank among the cell's survivors: by track order, or by (lifetime descending, track order) when the cell
    //      is cut.  New features have lifetime 1, survivors at least 2: the new ones always follow, in rank order.
    FB_FOR(k, n_tr) {
        const int c = L.b[k];
        const int m = cnt_t[c], total = m + new_cnt[c];
        const bool cut = total > B.grid_max;
        const int life = L.d[k], o = t_off[c];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const int k2 = L.tl[o + j];
            rank += cut ? ((L.d[k2] > life || (L.d[k2] == life && k2 < k)) ? 1 : 0) : (k2 < k ? 1 : 0);
        }
        if (rank >= B.grid_max) continue;
        const int at = out_off[c] + rank;
        if (at >= B.cap) { st.overflow = 1; continue; }
        B.curr.id[at] = B.tracked.id[k]; B.curr.lifetime[at] = life; B.curr.code[at] = c;
        B.curr.response[at] = B.tracked.response[k];
        B.curr.cam0[at] = B.tracked.cam0[k]; B.curr.cam1[at] = B.tracked.cam1[k];
        B.curr.und0[at] = B.tracked.und0[k]; B.curr.und1[at] = B.tracked.und1[k];
    }
    FB_FOR(i, n_cand) {
        const int r = L.a[i];
        if (r < 0) continue;
        const mskf_point2f p = B.cand_pt[i];
        const int c = fb_grid_code(B, p.x, p.y);
        const int slot = cnt_t[c] + r;
        if (slot >= B.grid_max) continue;
        const int at = out_off[c] + slot;
        if (at >= B.cap) { st.overflow = 1; continue; }
        B.curr.id[at] = id0 + (unsigned long long)(new_off[c] + r); B.curr.lifetime[at] = 1; B.curr.code[at] = c;
        B.curr.response[at] = (float)((double)L.c[i] / 256.0);
        B.curr.cam0[at] = B.c_out0[i]; B.curr.cam1[at] = B.c_out1[i];
        B.curr.und0[at] = B.c_und0[i]; B.curr.und1[at] = B.c_und1[i];
    }
    FB_SYNC();
    // ---- what the host gets (publish, :1137-1182, writes the message from it) and the state of the next frame
    const int n_out = n_curr < B.cap ? n_curr : B.cap;
    FB_FOR(o, n_out) {
        B.x_id[o] = B.curr.id[o]; B.x_lifetime[o] = B.curr.lifetime[o];
        B.x_cam0[o] = B.curr.cam0[o]; B.x_cam1[o] = B.curr.cam1[o];
        B.x_und0[o] = B.curr.und0[o]; B.x_und1[o] = B.curr.und1[o];
    }
    FB_FOR(one, 1) {
        st.n_new = n_new;
        st.n_curr = n_out;
        st.next_id = id0 + (unsigned long long)n_new;
        st.n_prev = n_out;
        B.x_info[0] = n_out; B.x_info[1] = n_cand; B.x_info[2] = st.before_tracking; B.x_info[3] = st.after_tracking;
        B.x_info[4] = st.after_matching; B.x_info[5] = st.after_ransac;
        const unsigned long long nid = id0 + (unsigned long long)n_new;
        B.x_info[6] = (int)(unsigned int)(nid & 0xFFFFFFFFULL); B.x_info[7] = (int)(unsigned int)(nid >> 32);
        B.x_info[8] = st.overflow; B.x_info[9] = n_new; B.x_info[10] = st.n_det; B.x_info[11] = n_tr;
        B.x_info[12] = (int)(unsigned int)(st.ransac_draws & 0xFFFFFFFFULL); B.x_info[13] = (int)(unsigned int)(st.ransac_draws >> 32);
    }
    FB_SYNC();
}
