/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the KGDet hot path.
 *
 * This library is the checker, not the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  kgdet_amd/ never imports it.
 *
 * Every function restates one piece of the reference (R = /root/reference/mmdetection/mmdet)
 * and cites the lines it follows.  Pinning status:
 *   - oracle_nms / oracle_soft_nms: PINNED against golden vectors produced by the compiled,
 *     unmodified reference sources (R/ops/nms/src/nms_cpu.cpp, soft_nms_cpu.pyx; see
 *     oracle/build_ref.py, tests/golden/make_nms_golden.py).
 *   - deformable conv v1/v2, deformable PS-RoI pooling, sigmoid focal loss: the reference has
 *     no CPU path and no tests for them ("parity unpinned" by the reference itself); they are
 *     cross-checked against independent formulations in tests/test_oracle_*.py
 *     (F.conv2d, F.grid_sample, autograd gradcheck, the reference's own pure-torch focal formula).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* deformable convolution column stages, float and double                                     */
/* ------------------------------------------------------------------------------------------ */
#define REAL float
#define SUFFIX f32
#define FABS fabsf
#include "dcn_oracle.inc"
#undef REAL
#undef SUFFIX
#undef FABS

#define REAL double
#define SUFFIX f64
#define FABS fabs
#include "dcn_oracle.inc"
#undef REAL
#undef SUFFIX
#undef FABS

/* ------------------------------------------------------------------------------------------ */
/* deformable PS-RoI pooling  (R/ops/dcn/src/deform_pool_cuda_kernel.cu)                      */
/* ------------------------------------------------------------------------------------------ */
#define REAL float
#define SUFFIX f32
#include "psroi_oracle.inc"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX f64
#include "psroi_oracle.inc"
#undef REAL
#undef SUFFIX

/* ------------------------------------------------------------------------------------------ */
/* hard NMS, CPU semantics  (R/ops/nms/src/nms_cpu.cpp:5-59)                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  float score;
  int64_t idx;
} scored_t;

static int by_score_desc_stable(const void *pa, const void *pb) {
  const scored_t *a = (const scored_t *)pa, *b = (const scored_t *)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx); /* ties: lower index first (stable) */
}

/*
 * dets [n,5] = x1,y1,x2,y2,score (float32).  Writes the kept box indices, ascending, to keep
 * and returns their count.  Visiting order = scores sorted descending (nms_cpu.cpp:20); the
 * reference's sort is not documented as stable, this oracle fixes ties to "lower index first".
 * Areas use the +1 convention (:18); a later box is suppressed when IoU >= thr (:55); the
 * result is nonzero(suppressed == 0), i.e. ascending index, NOT score order (:58).
 */
int64_t oracle_nms(const float *dets, int64_t n, float thr, int64_t *keep) {
  if (n == 0) return 0;
  scored_t *order = (scored_t *)malloc(sizeof(scored_t) * n);
  float *area = (float *)malloc(sizeof(float) * n);
  uint8_t *dead = (uint8_t *)calloc(n, 1);
  for (int64_t i = 0; i < n; ++i) {
    const float *d = dets + 5 * i;
    order[i].score = d[4];
    order[i].idx = i;
    area[i] = (d[2] - d[0] + 1) * (d[3] - d[1] + 1);
  }
  qsort(order, n, sizeof(scored_t), by_score_desc_stable);
  for (int64_t a = 0; a < n; ++a) {
    const int64_t i = order[a].idx;
    if (dead[i]) continue;
    const float *di = dets + 5 * i;
    for (int64_t b = a + 1; b < n; ++b) {
      const int64_t j = order[b].idx;
      if (dead[j]) continue;
      const float *dj = dets + 5 * j;
      const float xx1 = di[0] > dj[0] ? di[0] : dj[0];
      const float yy1 = di[1] > dj[1] ? di[1] : dj[1];
      const float xx2 = di[2] < dj[2] ? di[2] : dj[2];
      const float yy2 = di[3] < dj[3] ? di[3] : dj[3];
      float w = xx2 - xx1 + 1, h = yy2 - yy1 + 1;
      if (w < 0) w = 0;
      if (h < 0) h = 0;
      const float inter = w * h;
      const float ovr = inter / (area[i] + area[j] - inter);
      if (ovr >= thr) dead[j] = 1;
    }
  }
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (!dead[i]) keep[m++] = i;
  free(order);
  free(area);
  free(dead);
  return m;
}

/* ------------------------------------------------------------------------------------------ */
/* soft-NMS  (R/ops/nms/src/soft_nms_cpu.pyx:22-127)                                          */
/* ------------------------------------------------------------------------------------------ */
/*
 * boxes [n,5] float32 is modified in place (callers pass a copy); inds [n] receives the
 * original index of every surviving row.  Returns the surviving count N; rows [0,N) of
 * boxes/inds are the result.  method: 1 linear, 2 gaussian, other = hard.
 * Selection-sort structure, swap-with-last removal and the strict `iw > 0`, `ih > 0`,
 * `ov > iou_thr`, `score < min_score` tests follow the .pyx line for line; the gaussian
 * weight goes through double exp() exactly as numpy's np.exp on a Python float does (:107).
 */
int64_t oracle_soft_nms(float *boxes, int64_t n, float iou_thr, int method, float sigma,
                        float min_score, int64_t *inds) {
  int64_t N = n;
  for (int64_t i = 0; i < n; ++i) inds[i] = i;
  for (int64_t i = 0; i < N; ++i) {
    float maxscore = boxes[5 * i + 4];
    int64_t maxpos = i;
    float keep_row[5];
    memcpy(keep_row, boxes + 5 * i, sizeof keep_row);
    const int64_t keep_ind = inds[i];
    for (int64_t pos = i + 1; pos < N; ++pos)
      if (maxscore < boxes[5 * pos + 4]) {
        maxscore = boxes[5 * pos + 4];
        maxpos = pos;
      }
    memcpy(boxes + 5 * i, boxes + 5 * maxpos, sizeof keep_row);
    inds[i] = inds[maxpos];
    memcpy(boxes + 5 * maxpos, keep_row, sizeof keep_row);
    inds[maxpos] = keep_ind;

    const float tx1 = boxes[5 * i], ty1 = boxes[5 * i + 1];
    const float tx2 = boxes[5 * i + 2], ty2 = boxes[5 * i + 3];
    int64_t pos = i + 1;
    while (pos < N) {
      float *b = boxes + 5 * pos;
      const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
      /* cython turns the integer literal 1 into the double 1.0, so these promote (pyx:89-97) */
      const float area = (float)(((x2 - x1) + 1.0) * ((y2 - y1) + 1.0));
      const float iw = (float)(((tx2 <= x2 ? tx2 : x2) - (tx1 >= x1 ? tx1 : x1)) + 1.0);
      if (iw > 0.0) {
        const float ih = (float)(((ty2 <= y2 ? ty2 : y2) - (ty1 >= y1 ? ty1 : y1)) + 1.0);
        if (ih > 0.0) {
          const float ua =
              (float)(((((tx2 - tx1) + 1.0) * ((ty2 - ty1) + 1.0)) + area) - (iw * ih));
          const float ov = (iw * ih) / ua;
          float weight;
          if (method == 1)
            weight = ov > iou_thr ? (float)(1.0 - ov) : 1.0f;
          else if (method == 2)
            weight = (float)exp((double)((-(ov * ov)) / sigma));
          else
            weight = ov > iou_thr ? 0.0f : 1.0f;
          b[4] = weight * b[4];
          if (b[4] < min_score) {
            memcpy(b, boxes + 5 * (N - 1), sizeof keep_row);
            inds[pos] = inds[N - 1];
            N = N - 1;
            pos = pos - 1;
          }
        }
      }
      pos = pos + 1;
    }
  }
  return N;
}

/* ------------------------------------------------------------------------------------------ */
/* sigmoid focal loss  (R/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:24-97)        */
/* ------------------------------------------------------------------------------------------ */
/*
 * logits [num, C] float32, targets [num] int64 (0 = background, 1..C = class), losses [num, C].
 * The kernel's literals are double (`1.`, `2.`), so sub-expressions promote to double between
 * the float libm calls; the same promotions are written out here.
 */
static double focal_neg_log1mp(float x) { /* -x*[x>=0] - log(1 + exp(x - 2x*[x>=0]))  (:47-49) */
  const int ge = x >= 0;
  return -1. * x * ge - logf((float)(1. + expf((float)(x - 2. * x * ge))));
}

void oracle_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, int64_t num,
                                       int num_classes, float gamma, float alpha, float *losses) {
  for (int64_t i = 0; i < num * num_classes; ++i) {
    const int64_t n = i / num_classes;
    const int d = (int)(i % num_classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = (t >= 0 & t != (d + 1));
    const float zn = (float)(1.0 - alpha);
    const float zp = alpha;
    const float p = (float)(1. / (1. + expf(-logits[i])));
    const float term1 = powf((float)(1. - p), gamma) * logf(p > FLT_MIN ? p : FLT_MIN);
    const float term2 = (float)(powf(p, gamma) * focal_neg_log1mp(logits[i]));
    float l = 0.0f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

void oracle_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets,
                                        const float *d_losses, int64_t num, int num_classes,
                                        float gamma, float alpha, float *d_logits) {
  for (int64_t i = 0; i < num * num_classes; ++i) {
    const int64_t n = i / num_classes;
    const int d = (int)(i % num_classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = (t >= 0 & t != (d + 1));
    const float zn = (float)(1.0 - alpha);
    const float zp = alpha;
    const float p = (float)(1. / (1. + expf(-logits[i])));
    /* (1-p)^g * (1 - p - g*p*log(p))   (:80-81) */
    const float term1 = (float)(powf((float)(1. - p), gamma) *
                                (1. - p - (p * gamma * logf(p > FLT_MIN ? p : FLT_MIN))));
    /* p^g * (g*(1-p)*log(1-p) - p)     (:84-89) */
    const float term2 =
        (float)(powf(p, gamma) * (focal_neg_log1mp(logits[i]) * (1. - p) * gamma - p));
    float g = 0.0f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}
