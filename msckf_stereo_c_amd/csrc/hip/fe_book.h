// fe_book.h — the front-end's per-frame bookkeeping as device code: what cg::ImageProcessor does between and after the two
// track calls of a frame (reference msckf_core/src/image_processor.cpp, lines cited per step):
//
//   fe_book1 (after the temporal + stereo track of the previous features)
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
// host the items run one after the other.  No atomics, no cross-lane operations: sums and ranks come from scans with a
// fixed order, so the results do not depend on the schedule and the host run is a faithful check of the logic.
#pragma once
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
#else
#define FB_FOR(i, n) for (int i = 0; i < (n); ++i)
#define FB_SYNC() ((void)0)
#define FB_NTH 256
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
    int *x_info;                        // 16 ints: n_curr, n_cand, before/after tracking x4, next_id lo/hi, overflow
    unsigned long long *x_id;
    int *x_lifetime;
    mskf_point2f *x_cam0, *x_cam1, *x_und0, *x_und1;
};

// Everything a phase walks serially (one item per grid cell going over all survivors / detections / candidates) is staged
// here first: a serial walk over global memory pays a cache round trip per element (the first version of fe_book1 counted
// the survivors per cell that way and took 0.7 ms).
struct FeBookScratch {      // workgroup scratch (LDS on the device), carved from one int array by fe_book_scratch_init
    int *a;                 // max(cap, det_cap): flags / scan values / detection scores / ranked candidate lists
    int *b;                 // max(cap, det_cap): grid codes of the survivors / of the detections
    int *c;                 // cand_cap: per candidate, the score it is ranked with (-1: no stereo match)
    int *d;                 // cap: lifetimes of the survivors
    int *chunk;             // FB_NTH + 1: chunk sums of the scans
    int *cellA, *cellB, *cellC, *cellD;   // n_codes + 1 each
    unsigned char *occ;     // det_rows * det_cols
};

FB_FN size_t fe_book_scratch_ints(int cap, int cand_cap, int det_cap, int n_codes, int det_cells) {
    const int m = cap > det_cap ? cap : det_cap;
    return (size_t)2 * m + cand_cap + cap + (FB_NTH + 1) + (size_t)4 * (n_codes + 1) + (size_t)(det_cells + 3) / 4 + 8;
}
FB_FN void fe_book_scratch_init(FeBookScratch &L, int *mem, int cap, int cand_cap, int det_cap, int n_codes, int det_cells) {
    const int m = cap > det_cap ? cap : det_cap;
    L.a = mem; mem += m;
    L.b = mem; mem += m;
    L.c = mem; mem += cand_cap;
    L.d = mem; mem += cap;
    L.chunk = mem; mem += FB_NTH + 1;
    L.cellA = mem; mem += n_codes + 1;
    L.cellB = mem; mem += n_codes + 1;
    L.cellC = mem; mem += n_codes + 1;
    L.cellD = mem; mem += n_codes + 1;
    L.occ = (unsigned char *)mem;
    (void)det_cells;
}

// image_processor.cpp:452-454 (Q7: the column may equal grid_col): float division, truncation
FB_FN int fb_grid_code(const FeBookDev &B, float x, float y) {
    return (int)(y / (float)B.grid_h) * B.grid_col + (int)(x / (float)B.grid_w);
}

// Exclusive prefix sum of v[0, n) in place, total returned through *total (every caller passes workgroup-shared memory).
// Chunks of consecutive items are summed by one item each, the chunk sums are scanned serially by one item, the chunks are
// then rewritten: fixed order, no atomics.
FB_FN void fb_exclusive_scan(int *v, int n, int *chunk, int *total) {
    const int nth = FB_NTH;
    const int per = (n + nth - 1) / nth;
    const int nch = per > 0 ? (n + per - 1) / per : 0;
    FB_FOR(c, nch) {
        int s = 0;
        const int e = (c + 1) * per < n ? (c + 1) * per : n;
        for (int i = c * per; i < e; ++i) s += v[i];
        chunk[c] = s;
    }
    FB_SYNC();
    FB_FOR(one, 1) {
        int run = 0;
        for (int c = 0; c < nch; ++c) { const int t = chunk[c]; chunk[c] = run; run += t; }
        *total = run;
    }
    FB_SYNC();
    FB_FOR(c, nch) {
        int run = chunk[c];
        const int e = (c + 1) * per < n ? (c + 1) * per : n;
        for (int i = c * per; i < e; ++i) { const int t = v[i]; v[i] = run; run += t; }
    }
    FB_SYNC();
}

// ------------------------------------------------------------------------------------------ after the first track call
FB_FN void fe_book1(const FeBookDev &B, FeBookScratch &L) {
    FeBookState &st = *B.st;
    const int n = st.n_prev;
    const int det_cells = B.det_rows * B.det_cols;
    int *tot = L.chunk + FB_NTH;         // scan totals land here (shared)
    // ---- trackFeatures tail (:416-513).  status bit 0: temporal track inside the image, bit 1: stereo match accepted (only
    //      ever set together with bit 0).  Survivors keep their order (:440, :465-480 compaction), lifetime + 1 (:508).
    FB_FOR(i, n) L.a[i] = (B.t_status[i] & 3) == 3 ? 1 : 0;
    FB_FOR(c, B.n_codes + 1) { L.cellA[c] = 0; }
    FB_FOR(k, det_cells) L.occ[k] = 0;
    FB_SYNC();
    // tracking info: features with bit 0 (the scan of a copy in b gives the count)
    FB_FOR(i, n) L.b[i] = (B.t_status[i] & 1) ? 1 : 0;
    FB_SYNC();
    fb_exclusive_scan(L.b, n, L.chunk, tot);
    const int n_bit0 = *tot;
    FB_SYNC();
    fb_exclusive_scan(L.a, n, L.chunk, tot);
    const int n_tr = *tot;
    FB_SYNC();
    FB_FOR(i, n) {
        if ((B.t_status[i] & 3) != 3) continue;
        const int k = L.a[i];
        const mskf_point2f p = B.t_out0[i];
        int code = fb_grid_code(B, p.x, p.y);
        if (code < 0) code = 0;
        if (code >= B.n_codes) code = B.n_codes - 1;      // (cannot happen for a point inside the image: n_codes covers every code)
        B.tracked.id[k] = B.prev.id[i];
        B.tracked.lifetime[k] = B.prev.lifetime[i] + 1;
        B.tracked.code[k] = code;
        L.b[k] = code;                                      // (b held the bit-0 flags' scan: dead since n_bit0 was read)
        B.tracked.response[k] = 0.f;
        B.tracked.cam0[k] = p; B.tracked.cam1[k] = B.t_out1[i];
        B.tracked.und0[k] = B.t_und0[i]; B.tracked.und1[k] = B.t_und1[i];
        // CornerDetector::set_grid_position of the truncated pixel (:632-649)
        const int xi = (int)p.x, yi = (int)p.y;
        int r = (int)((float)yi / (float)B.det_ch), c = (int)((float)xi / (float)B.det_cw);
        r = r < 0 ? 0 : (r >= B.det_rows ? B.det_rows - 1 : r);
        c = c < 0 ? 0 : (c >= B.det_cols ? B.det_cols - 1 : c);
        L.occ[r * B.det_cols + c] = 1;                      // (several items may store the same 1)
    }
    FB_SYNC();
    // tracked features per grid code: one item per code walks the survivors (no atomics)
    FB_FOR(c, B.n_codes) {
        int cnt = 0;
        for (int k = 0; k < n_tr; ++k) cnt += L.b[k] == c ? 1 : 0;
        L.cellA[c] = cnt;
        B.cell_count[c] = cnt;
    }
    FB_FOR(one, 1) {
        st.n_tracked = n_tr;
        st.before_tracking = n;
        if (n > 0) { st.after_tracking = n_bit0; st.after_matching = n_tr; st.after_ransac = n_tr; }   // (:383: nothing is touched without features)
    }
    FB_SYNC();
    // ---- detections (:657): cells in order whose maximum beats the threshold and that hold no live feature
    FB_FOR(k, det_cells) {
        const unsigned long long key = B.cell_keys[k];
        const int score = (unsigned int)(key >> 56) == B.gen ? (int)((key >> 32) & 0xFFFFFFULL) : 0;
        L.a[k] = (score > B.thr_score && !L.occ[k]) ? 1 : 0;
    }
    FB_SYNC();
    FB_FOR(k, det_cells) L.b[k] = L.a[k];
    FB_SYNC();
    fb_exclusive_scan(L.a, det_cells, L.chunk, tot);
    const int n_det = *tot;
    FB_SYNC();
    FB_FOR(k, det_cells) {
        if (!L.b[k]) continue;
        const unsigned long long key = B.cell_keys[k];
        const unsigned int order = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFULL);
        const int cy = k / B.det_cols, cx = k - cy * B.det_cols;
        const int q = L.a[k];
        mskf_point2f p;
        p.y = (float)(cy * B.det_ch + (int)(order / (unsigned)B.det_cw));
        p.x = (float)(cx * B.det_cw + (int)(order % (unsigned)B.det_cw));
        B.det_pt[q] = p;
        B.det_score[q] = (int)((key >> 32) & 0xFFFFFFULL);
    }
    FB_SYNC();
    // grid code and score of every detection (walked by every cell below)
    FB_FOR(q, n_det) { const mskf_point2f p = B.det_pt[q]; L.b[q] = fb_grid_code(B, p.x, p.y); L.a[q] = B.det_score[q]; }
    FB_SYNC();
    // ---- sieve (:661-677): every grid cell keeps its grid_max best detections by response, stable (equal responses keep
    //      their detection order).  Only the cells with a vacancy send theirs on (a full cell's candidates cannot influence any
    //      output), but every cell's kept count moves the position in the reference's full candidate list (Q4).
    //      cellB = kept count of every cell, cellC = candidates of the cell (0 for a full one)
    FB_FOR(c, B.n_cells) {
        int cnt = 0;
        for (int q = 0; q < n_det; ++q) cnt += L.b[q] == c ? 1 : 0;
        const int kept = cnt < B.grid_max ? cnt : B.grid_max;
        L.cellB[c] = kept;
        L.cellC[c] = L.cellA[c] < B.grid_min ? kept : 0;
    }
    FB_SYNC();
    FB_FOR(c, B.n_cells) L.cellD[c] = L.cellB[c];
    FB_SYNC();
    fb_exclusive_scan(L.cellD, B.n_cells, L.chunk, tot);          // cellD = position of the cell's first candidate in the full list
    FB_SYNC();
    FB_FOR(c, B.n_cells) B.cand_cnt[c] = L.cellC[c];
    FB_SYNC();
    fb_exclusive_scan(L.cellC, B.n_cells, L.chunk, tot);          // cellC = offset of the cell in the list that is sent on
    const int n_cand = *tot;
    FB_SYNC();
    FB_FOR(c, B.n_cells) {
        B.cand_off[c] = L.cellC[c];
        if (B.cand_cnt[c] <= 0) continue;
        // A cell with more than grid_max detections keeps its best grid_max, descending response, ties in detection order
        // (insertion into a short list); a cell with fewer is NOT sorted (:664 sorts only when it has to cut): detection order.
        int best_q[FB_MAXK], best_s[FB_MAXK];
        int m = 0, cnt = 0;
        const int K = B.grid_max;
        for (int q = 0; q < n_det; ++q) cnt += L.b[q] == c ? 1 : 0;
        const bool cut = cnt > K;
        for (int q = 0; q < n_det; ++q) {
            if (L.b[q] != c) continue;
            const int s = L.a[q];
            if (!cut) { best_q[m] = q; best_s[m] = s; ++m; continue; }
            if (m == K && !(s > best_s[K - 1])) continue;
            int pos = m < K ? m : K - 1;
            while (pos > 0 && s > best_s[pos - 1]) { best_q[pos] = best_q[pos - 1]; best_s[pos] = best_s[pos - 1]; --pos; }
            best_q[pos] = q; best_s[pos] = s;
            if (m < K) ++m;
        }
        const int off = L.cellC[c], flat = L.cellD[c];
        for (int k = 0; k < m; ++k) {
            if (off + k >= B.cand_cap) { st.overflow = 1; break; }
            B.cand_pt[off + k] = B.det_pt[best_q[k]];
            B.cand_score[off + k] = best_s[k];
            B.cand_index[off + k] = flat + k;
        }
    }
    FB_FOR(one, 1) { st.n_det = n_det; st.n_cand = n_cand < B.cand_cap ? n_cand : B.cand_cap; }
    FB_SYNC();
}

// ------------------------------------------------------------------------------------------ after the second track call
FB_FN void fe_book2(const FeBookDev &B, FeBookScratch &L) {
    FeBookState &st = *B.st;
    const int n_tr = st.n_tracked, n_cand = st.n_cand, n_det = st.n_det;
    int *tot = L.chunk + FB_NTH;
    // codes and lifetimes of the survivors into scratch (every cell walks them), and per candidate the score it is ranked
    // with: under Q4 the detection-order score at the candidate's position in the full candidate list (:698), else its own
    FB_FOR(k, n_tr) { L.b[k] = B.tracked.code[k]; L.d[k] = B.tracked.lifetime[k]; }
    FB_FOR(i, n_cand) {
        int sc = -1;
        if (B.c_status[i] & 2) {
            sc = B.cand_score[i];
            if (B.q4) { const int di = B.cand_index[i]; sc = di < n_det ? B.det_score[di] : 0; }
        }
        L.c[i] = sc;
    }
    FB_FOR(c, B.n_codes + 1) { L.cellA[c] = c < B.n_codes ? B.cell_count[c] : 0; }
    FB_SYNC();
    // ---- addNewFeatures tail (:700-750): per cell, the matched candidates ranked by response (stable) fill the vacancy.
    //      cellB = new features of the cell.  The response a candidate is ranked with is, under Q4, the detection-order
    //      response at the candidate's position in the full candidate list (:698), else its own.
    //      The ranked list of a cell is kept in a[c * grid_min ..] for the emit phase (candidate indices; cap >= n_codes * grid_max).
    FB_FOR(c, B.n_codes) {
        int m = 0;
        if (c < B.n_cells) {
            const int vac = B.grid_min - L.cellA[c];
            const int K = vac < FB_MAXK ? vac : FB_MAXK;
            if (K > 0) {
                int best_i[FB_MAXK];
                float best_r[FB_MAXK];
                const int o = B.cand_off[c], e = o + B.cand_cnt[c];
                for (int i = o; i < e && i < n_cand; ++i) {
                    const int sc = L.c[i];
                    if (sc < 0) continue;
                    const float r = (float)((double)sc / 256.0);
                    if (m == K && !(r > best_r[K - 1])) continue;
                    int pos = m < K ? m : K - 1;
                    while (pos > 0 && r > best_r[pos - 1]) { best_i[pos] = best_i[pos - 1]; best_r[pos] = best_r[pos - 1]; --pos; }
                    best_i[pos] = i; best_r[pos] = r;
                    if (m < K) ++m;
                }
                for (int k = 0; k < m; ++k) L.a[c * B.grid_min + k] = best_i[k];
            }
        }
        L.cellB[c] = m;
    }
    FB_SYNC();
    FB_FOR(c, B.n_codes) L.cellC[c] = L.cellB[c];
    FB_SYNC();
    fb_exclusive_scan(L.cellC, B.n_codes, L.chunk, tot);          // cellC = rank of the cell's first new feature: ids in ascending cell order (:745)
    const int n_new = *tot;
    FB_SYNC();
    // ---- this frame's grid (:498-513, :735-750) and pruneGridFeatures (:758-768): per cell the survivors in track order,
    //      then the new features in rank order; a cell over grid_max keeps its grid_max longest-lived, stable.
    //      cellD = features the cell publishes
    FB_FOR(c, B.n_codes) {
        const int total = L.cellA[c] + L.cellB[c];
        L.cellD[c] = total < B.grid_max ? total : B.grid_max;
    }
    FB_SYNC();
    // (the scan overwrites cellD with the offsets; the counts are recomputed where they are needed)
    fb_exclusive_scan(L.cellD, B.n_codes, L.chunk, tot);
    const int n_curr = *tot;
    FB_SYNC();
    const unsigned long long id0 = st.next_id;
    FB_FOR(c, B.n_codes) {
        const int n_t = L.cellA[c], n_n = L.cellB[c], total = n_t + n_n;
        const int out0 = L.cellD[c];
        if (total == 0) continue;
        // member m of the cell: m < n_t -> the m-th survivor with this code (track order); else new feature m - n_t
        auto emit = [&](int slot, int src_tracked, int k) {
            const int o = out0 + slot;
            if (o >= B.cap) { st.overflow = 1; return; }
            if (src_tracked) {
                B.curr.id[o] = B.tracked.id[k]; B.curr.lifetime[o] = B.tracked.lifetime[k]; B.curr.code[o] = c;
                B.curr.response[o] = B.tracked.response[k];
                B.curr.cam0[o] = B.tracked.cam0[k]; B.curr.cam1[o] = B.tracked.cam1[k];
                B.curr.und0[o] = B.tracked.und0[k]; B.curr.und1[o] = B.tracked.und1[k];
            } else {
                const int i = L.a[c * B.grid_min + k];
                const int sc = L.c[i];
                B.curr.id[o] = id0 + (unsigned long long)(L.cellC[c] + k); B.curr.lifetime[o] = 1; B.curr.code[o] = c;
                B.curr.response[o] = (float)((double)sc / 256.0);
                B.curr.cam0[o] = B.c_out0[i]; B.curr.cam1[o] = B.c_out1[i];
                B.curr.und0[o] = B.c_und0[i]; B.curr.und1[o] = B.c_und1[i];
            }
        };
        if (total <= B.grid_max) {
            int slot = 0;
            for (int k = 0; k < n_tr && slot < n_t; ++k) if (L.b[k] == c) emit(slot++, 1, k);
            for (int k = 0; k < n_n; ++k) emit(n_t + k, 0, k);
        } else {
            // the grid_max longest-lived of the cell's members, ties in member order: insertion into a short list
            int best_src[FB_MAXK], best_k[FB_MAXK], best_life[FB_MAXK];
            int m = 0;
            const int K = B.grid_max;
            auto offer = [&](int src_tracked, int k, int life) {
                if (m == K && !(life > best_life[K - 1])) return;
                int pos = m < K ? m : K - 1;
                while (pos > 0 && life > best_life[pos - 1]) { best_src[pos] = best_src[pos - 1]; best_k[pos] = best_k[pos - 1]; best_life[pos] = best_life[pos - 1]; --pos; }
                best_src[pos] = src_tracked; best_k[pos] = k; best_life[pos] = life;
                if (m < K) ++m;
            };
            for (int k = 0; k < n_tr; ++k) if (L.b[k] == c) offer(1, k, L.d[k]);
            for (int k = 0; k < n_n; ++k) offer(0, k, 1);
            for (int s = 0; s < m; ++s) emit(s, best_src[s], best_k[s]);
        }
    }
    FB_SYNC();
    // ---- what the host gets (publish, :1137-1182, writes the message from it) and the state of the next frame
    const int n_out = n_curr < B.cap ? n_curr : B.cap;
    FB_FOR(o, n_out) {
        B.x_id[o] = B.curr.id[o]; B.x_lifetime[o] = B.curr.lifetime[o];
        B.x_cam0[o] = B.curr.cam0[o]; B.x_cam1[o] = B.curr.cam1[o];
        B.x_und0[o] = B.curr.und0[o]; B.x_und1[o] = B.curr.und1[o];
    }
    FB_SYNC();
    FB_FOR(one, 1) {
        st.n_new = n_new;
        st.n_curr = n_out;
        st.next_id = id0 + (unsigned long long)n_new;
        st.n_prev = n_out;
        B.x_info[0] = n_out; B.x_info[1] = n_cand; B.x_info[2] = st.before_tracking; B.x_info[3] = st.after_tracking;
        B.x_info[4] = st.after_matching; B.x_info[5] = st.after_ransac;
        B.x_info[6] = (int)(unsigned int)(st.next_id & 0xFFFFFFFFULL); B.x_info[7] = (int)(unsigned int)(st.next_id >> 32);
        B.x_info[8] = st.overflow; B.x_info[9] = n_new; B.x_info[10] = st.n_det; B.x_info[11] = n_tr;
    }
    FB_SYNC();
}
