/*
 * TEST INFRASTRUCTURE - CPU oracle of the post-processing half of the path.  Not product code.
 *
 * Plain C restatement of
 *   - torchvision.ops.boxes.nms (torchvision>=0.6, requirements.txt:10; the single call site is
 *     utils/utils.py:627).  torchvision is neither shipped with the reference nor installed here, so
 *     this follows its published CPU kernel: areas = (x2-x1)*(y2-y1); visit boxes in stable descending
 *     score order; a visited, not yet suppressed box i is kept and suppresses every later box j with
 *     inter / (area_i + area_j - inter) > iou_threshold, inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1)),
 *     fp32 arithmetic, comparison against the double threshold.  No reference test or fixture pins this
 *     primitive ("parity unpinned"); tests cross-check it against an independent numpy restatement.
 *   - the candidate stage list of non_max_suppression for merge=False (utils/utils.py:570-629, 641).
 * Built by oracle/Makefile into oracle/_build/libnmsref.so; compile without FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int64_t idx; } item_t;

/* stable merge sort, descending score */
static void msort(item_t* a, item_t* tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    msort(a, tmp, h);
    msort(a + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].score > a[i].score) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(item_t));
}

/* keep receives up to max_keep indices (descending score); returns the TOTAL number kept (may exceed max_keep),
 * or -1 when out of memory. Stops early once max_keep boxes are kept when stop_at_max != 0. */
int64_t y5ref_nms(const float* boxes, const float* scores, int64_t n, double thr, int64_t* keep, int64_t max_keep,
                  int stop_at_max) {
    if (n <= 0) return 0;
    item_t* it = (item_t*)malloc((size_t)n * sizeof(item_t));
    item_t* tmp = (item_t*)malloc((size_t)n * sizeof(item_t));
    float* area = (float*)malloc((size_t)n * sizeof(float));
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    if (!it || !tmp || !area || !dead) { free(it); free(tmp); free(area); free(dead); return -1; }
    for (int64_t i = 0; i < n; ++i) { it[i].score = scores[i]; it[i].idx = i; }
    msort(it, tmp, n);
    for (int64_t i = 0; i < n; ++i) {
        const float* b = boxes + 4 * i;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    int64_t kept = 0;
    for (int64_t a = 0; a < n; ++a) {
        const int64_t i = it[a].idx;
        if (dead[i]) continue;
        if (kept < max_keep) keep[kept] = i;
        ++kept;
        if (stop_at_max && kept >= max_keep) break;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = area[i];
        for (int64_t c = a + 1; c < n; ++c) {
            const int64_t j = it[c].idx;
            if (dead[j]) continue;
            const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (iarea + area[j] - inter);
            if ((double)ovr > thr) dead[j] = 1;
        }
    }
    free(it); free(tmp); free(area); free(dead);
    return kept;
}

/*
 * non_max_suppression for ONE image, merge=False.
 * pred: (n_rows, no) float32 rows [cx, cy, w, h, obj, cls...] (fp16 predictions are widened by the caller, :570-571).
 * class_keep: nc bytes or NULL.  det: (max_det, 6); kept_index: (max_det) flat candidate index row*nc + cls.
 * Returns the number of detections written (<= max_det), -1 when out of memory; *n_candidates receives the
 * candidate count after the confidence/class filters.
 */
int64_t y5ref_nms_image(const float* pred, int64_t n_rows, int64_t no, float conf_thres, double iou_thres,
                        const unsigned char* class_keep, int agnostic, int64_t max_det, float* det,
                        int64_t* kept_index, int64_t* n_candidates) {
    const int64_t nc = no - 5;
    int64_t cap = 1024, n = 0;
    float* cb = (float*)malloc((size_t)cap * 4 * sizeof(float));   /* class-offset boxes */
    float* ob = (float*)malloc((size_t)cap * 4 * sizeof(float));   /* plain boxes */
    float* sc = (float*)malloc((size_t)cap * sizeof(float));
    int64_t* fl = (int64_t*)malloc((size_t)cap * sizeof(int64_t));
    if (!cb || !ob || !sc || !fl) { free(cb); free(ob); free(sc); free(fl); return -1; }
    for (int64_t r = 0; r < n_rows; ++r) {
        const float* p = pred + r * no;
        const float obj = p[4];
        if (!(obj > conf_thres)) continue;                                   /* :574, :588 */
        const float x1 = p[0] - p[2] / 2, y1 = p[1] - p[3] / 2;             /* :598 xywh2xyxy */
        const float x2 = p[0] + p[2] / 2, y2 = p[1] + p[3] / 2;
        int64_t c_lo = 0, c_hi = nc;
        float best = 0.f;
        if (nc == 1) {                                                       /* :604-606 best class only */
            best = p[5] * obj;
            if (!(best > conf_thres)) continue;
        }
        for (int64_t c = c_lo; c < c_hi; ++c) {
            const float conf = (nc == 1) ? best : p[5 + c] * obj;            /* :595 conf = obj * cls */
            if (!(conf > conf_thres)) continue;                              /* :602 */
            if (class_keep && !class_keep[c]) continue;                      /* :609-610 */
            if (n == cap) {
                cap *= 2;
                cb = (float*)realloc(cb, (size_t)cap * 4 * sizeof(float));
                ob = (float*)realloc(ob, (size_t)cap * 4 * sizeof(float));
                sc = (float*)realloc(sc, (size_t)cap * sizeof(float));
                fl = (int64_t*)realloc(fl, (size_t)cap * sizeof(int64_t));
                if (!cb || !ob || !sc || !fl) { free(cb); free(ob); free(sc); free(fl); return -1; }
            }
            const float off = agnostic ? 0.f : (float)c * 4096.f;            /* :625 */
            ob[4 * n] = x1; ob[4 * n + 1] = y1; ob[4 * n + 2] = x2; ob[4 * n + 3] = y2;
            cb[4 * n] = x1 + off; cb[4 * n + 1] = y1 + off; cb[4 * n + 2] = x2 + off; cb[4 * n + 3] = y2 + off;  /* :626 */
            sc[n] = conf;
            fl[n] = r * nc + c;
            ++n;
        }
    }
    if (n_candidates) *n_candidates = n;
    int64_t* keep = (int64_t*)malloc((size_t)(max_det > 0 ? max_det : 1) * sizeof(int64_t));
    if (!keep) { free(cb); free(ob); free(sc); free(fl); return -1; }
    int64_t k = y5ref_nms(cb, sc, n, iou_thres, keep, max_det, 1);           /* :627-629 */
    if (k > max_det) k = max_det;
    for (int64_t i = 0; i < k; ++i) {
        const int64_t j = keep[i];
        det[6 * i] = ob[4 * j]; det[6 * i + 1] = ob[4 * j + 1]; det[6 * i + 2] = ob[4 * j + 2]; det[6 * i + 3] = ob[4 * j + 3];
        det[6 * i + 4] = sc[j];
        det[6 * i + 5] = (float)(fl[j] % nc);
        if (kept_index) kept_index[i] = fl[j];
    }
    free(cb); free(ob); free(sc); free(fl); free(keep);
    return k;
}
