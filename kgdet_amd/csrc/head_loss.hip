// Target assignment + the nine losses of the KGDet head as four launches (gfx950).
//
// What it replaces (one pyramid level -- KGDet: stride 32, 25 x 42 = 1050 points per image):
//   PointAssigner.assign ................. mmdet/core/bbox/assigners/point_assigner.py:23-121
//   point_target_kp / _single ............ mmdet/core/anchor/point_target_kp.py:7-169
//   offset_to_pts, loss_single, loss ..... mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:537-665
//   FocalLoss / SmoothL1Loss reductions .. mmdet/models/losses/{focal_loss.py:28-82, smooth_l1_loss.py:8-45, utils.py:7-52},
//                                          ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:24-97
// In torch that is ~650 small launches per step (topk / scatter / where chains per ground truth, six permute + flip +
// decode chains, target tensors of 2100 x 588 built with where(), weight normalisation, the focal weight / sum tail and
// the backward of every one of those nodes): 2.5 ms of a 14.6 ms step, all launch latency.  Here:
//
//   head_assign_select   block (gt, image): distance of every point to the gt centre (normalised by the gt size), the
//                        pos_num nearest by RANK among the ~2 pos_num candidates inside a bounding distance (ties by
//                        point index).
//   head_loss_forward    block (64-point tile, channel group, image): final assignment of its points (sequential over
//                        the gts: `min_dist < assigned_dist`, earlier gt wins ties), then the loss ROWS of its group --
//                        a row = one channel of one prediction map, lanes = points (coalesced NCHW reads): focal
//                        (13 x 3 rows), smooth-L1 boxes (4 x 3) and keypoints (588 x 3) on coordinates decoded in
//                        registers (pred * stride + centre, (y, x) -> (x, y)), targets gathered from the gt tables by
//                        the assigned index, keypoint weights 4 / (2 n_visible).  Nine per-workgroup partial sums.
//   head_loss_finish     one block: positives per image, num_total = sum max(n_pos, 1), partials in fixed order,
//                        loss_k = weight_k * (sum_k / num_total).
//   head_loss_backward   same rows: grad of every prediction map, written in full (zeros where the reference's weight is 0).
//
// No tensor of targets or weights exists; nothing is read by the host.  Deterministic (fixed summation orders).
#include <float.h>

#include "common.h"

namespace kgdet {

namespace {

constexpr int kMaxImages = KGDET_HEAD_MAX_IMAGES;
constexpr int kMaxGt = 64;
constexpr int kMaxPoints = 4096;

__device__ __forceinline__ double neg_softplus_d(float x) {  // as csrc/focal.hip
  const int ge = x >= 0;
  return -1. * x * ge - logf((float)(1. + expf((float)(x - 2. * x * ge))));
}

// sigmoid_focal_loss_cuda.cu:24-59 (same promotions as csrc/focal.hip)
__device__ __forceinline__ float focal_fwd(float x, int t, int d, float gamma, float alpha) {
  const float c1 = (t == (d + 1));
  const float c2 = ((t >= 0) & (t != (d + 1)));
  const float zn = (float)(1.0 - alpha), zp = alpha;
  const float p = (float)(1. / (1. + expf(-x)));
  const float term1 = powf((float)(1. - p), gamma) * logf(fmaxf(p, FLT_MIN));
  const float term2 = (float)(powf(p, gamma) * neg_softplus_d(x));
  float l = 0.0f;
  l += -c1 * term1 * zp;
  l += -c2 * term2 * zn;
  return l;
}

// :62-97
__device__ __forceinline__ float focal_bwd(float x, int t, int d, float gamma, float alpha) {
  const float c1 = (t == (d + 1));
  const float c2 = ((t >= 0) & (t != (d + 1)));
  const float zn = (float)(1.0 - alpha), zp = alpha;
  const float p = (float)(1. / (1. + expf(-x)));
  const float term1 = (float)(powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf(fmaxf(p, FLT_MIN)))));
  const float term2 = (float)(powf(p, gamma) * (neg_softplus_d(x) * (1. - p) * gamma - p));
  float g = 0.0f;
  g += -c1 * term1 * zp;
  g += -c2 * term2 * zn;
  return g;
}

// the assignment of point i of image b from the per-gt selections: point_assigner.py:106-117 (`min_dist <
// assigned_gt_dist[point_index]`, gts in order: the earlier gt keeps a tie)
__device__ __forceinline__ int final_assign(const float *__restrict__ dsel, int b, int gmax, int n_gt, int N, int i) {
  float best = INFINITY;
  int a = 0;
  for (int g = 0; g < n_gt; ++g) {
    const float d = dsel[((long long)b * gmax + g) * N + i];
    if (d < best) { best = d; a = g + 1; }
  }
  return a;
}

}  // namespace

// block (g, b), 256 threads.  dsel[b][g][i] = distance if point i is among the pos_num nearest of gt g, else +inf.
// "Among the pos_num nearest" = rank < pos_num under the (distance, point index) order.  Ranks are only needed for the
// points within T = the largest distance inside a ceil(sqrt(pos_num))-sided block of grid points around the centre: that
// block holds >= pos_num points, so every point beyond T has rank >= pos_num, and every point that precedes a candidate
// is a candidate itself -- the all-pairs count runs over ~2 pos_num candidates instead of all H * W points.
__global__ __launch_bounds__(256) void head_assign_select(const kgdet_head_targets t, int pos_num, int gmax,
                                                          float *__restrict__ dsel) {
  extern __shared__ float dist[];                    // [N] distances, then [N] candidate indices
  __shared__ int s_tbits, s_count;
  const int g = blockIdx.x, b = blockIdx.y, N = t.H * t.W, tid = threadIdx.x;
  if (g >= t.num_gt[b]) return;
  int *cand = reinterpret_cast<int *>(dist + N);
  const float *box = t.gt_bboxes[b] + 4 * g;
  // point_assigner.py:69-71: centre and size of the gt; :93-95: ((p - centre) / size).norm(dim=1)
  const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  const float w = fmaxf(box[2] - box[0], 1e-6f), h = fmaxf(box[3] - box[1], 1e-6f);
  if (tid == 0) { s_tbits = 0; s_count = 0; }
  // the image's valid extent (point_target_kp.py:107-118: the assigner only sees the points inside it)
  const int vh = t.valid_h[b] > 0 ? min(t.valid_h[b], t.H) : t.H, vw = t.valid_w[b] > 0 ? min(t.valid_w[b], t.W) : t.W;
  for (int i = tid; i < N; i += 256) {
    const int col = i % t.W, row = i / t.W;
    const float px = (float)col * t.stride, py = (float)row * t.stride;
    const float dx = (px - cx) / w, dy = (py - cy) / h;
    dist[i] = (row < vh && col < vw) ? sqrtf(dx * dx + dy * dy) : INFINITY;
  }
  __syncthreads();
  int side = 1;
  while (side * side < pos_num) ++side;
  float T = FLT_MAX;           // (every valid point is a candidate; invalid ones -- distance +inf -- never are)
  if (side <= vh && side <= vw) {
    const int j0 = min(max((int)floorf(cx / t.stride + 0.5f) - side / 2, 0), vw - side);
    const int i0 = min(max((int)floorf(cy / t.stride + 0.5f) - side / 2, 0), vh - side);
    for (int e = tid; e < side * side; e += 256)
      atomicMax(&s_tbits, __float_as_int(dist[(i0 + e / side) * t.W + j0 + e % side]));   // (distances are >= 0)
    __syncthreads();
    T = __int_as_float(s_tbits);
  }
  for (int i = tid; i < N; i += 256)
    if (dist[i] <= T) cand[atomicAdd(&s_count, 1)] = i;       // (any order: only counts are taken over the list)
  __syncthreads();
  const int M = s_count;
  float *out = dsel + ((long long)b * gmax + g) * N;
  for (int i = tid; i < N; i += 256) out[i] = INFINITY;
  __syncthreads();
  for (int c = tid; c < M; c += 256) {
    const int i = cand[c];
    const float di = dist[i];
    int rank = 0;
    for (int e = 0; e < M; ++e) {
      const int j = cand[e];
      const float dj = dist[j];
      rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
    }
    if (rank < pos_num) out[i] = di;
  }
}

namespace {

// Row enumeration shared by forward and backward: row r of [0, 3 * (C + 4 + 2 K)) -> (stage, kind, channel).
struct Row {
  int stage, kind, c;   // kind 0: cls, 1: bbox, 2: keypoints
};
__device__ __forceinline__ Row row_of(int r, int C, int K2) {
  const int per = C + 4 + K2;
  Row q;
  q.stage = r / per;
  int c = r - q.stage * per;
  if (c < C) { q.kind = 0; q.c = c; }
  else if (c < C + 4) { q.kind = 1; q.c = c - C; }
  else { q.kind = 2; q.c = c - C - 4; }
  return q;
}

struct PointCtx {
  int a, label, nvis;      // assigned gt (0: none), its label, its number of visible keypoints
  float px, py;
};

}  // namespace

// block (tile of 64 points, channel group, image); 256 threads = 4 waves, lane = point, waves take rows round-robin.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void head_loss_rows(const kgdet_head_targets t, const kgdet_head_loss_cfg cfg,
                                                      const kgdet_head_maps maps, const float *__restrict__ dsel, int gmax,
                                                      int rows_per_group, float *__restrict__ partial,
                                                      const float *__restrict__ num_total_dev,
                                                      const float *__restrict__ upstream, kgdet_head_maps grads) {
  __shared__ int s_nvis[kMaxGt];
  __shared__ float s_part[4][9];
  const int N = t.H * t.W, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < N;
  const int ic = min(i, N - 1);
  const int n_gt = t.num_gt[b];
  const int K = t.num_keypoints, K2 = 2 * K, C = t.num_classes;
  // visible keypoints per gt (kpt_weights.sum(1) / 2 of KP3:641-644)
  for (int g = wave; g < n_gt; g += 4) {
    int cnt = 0;
    for (int m = lane; m < K; m += 64) cnt += t.gt_keypoints[b][((long long)g * K + m) * 3 + 2] != 0.0f ? 1 : 0;
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) s_nvis[g] = cnt;
  }
  __syncthreads();
  PointCtx p;
  p.a = final_assign(dsel, b, gmax, n_gt, N, ic);
  p.label = p.a > 0 ? (t.gt_labels[b] ? (int)t.gt_labels[b][p.a - 1] : 1) : 0;
  p.nvis = p.a > 0 ? s_nvis[p.a - 1] : 0;
  p.px = (float)(ic % t.W) * t.stride;
  p.py = (float)(ic / t.W) * t.stride;
  const float *gbox = t.gt_bboxes[b] + 4 * max(p.a - 1, 0);
  const float *gkp = t.gt_keypoints[b] + (long long)max(p.a - 1, 0) * K * 3;
  const int vh = t.valid_h[b] > 0 ? min(t.valid_h[b], t.H) : t.H, vw = t.valid_w[b] > 0 ? min(t.valid_w[b], t.W) : t.W;
  const bool inside = ic / t.W < vh && ic % t.W < vw;                // (a point outside is never assigned: p.a == 0)
  const float label_w = p.a > 0 ? cfg.pos_weight : (inside ? 1.0f : 0.0f);   // point_target_kp.py:140-147, unmap fill 0
  const float kp_w = p.nvis > 0 ? 1.0f / (float)(2 * p.nvis) * 4.0f : 0.0f;
  const float nt = cfg.normalize_term;

  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  float inv_total = 0.f;
  if (BACKWARD) inv_total = 1.0f;   // (divided below: the reference divides the sum, then multiplies by loss_weight)

  const int total_rows = 3 * (C + 4 + K2);
  const int r_begin = blockIdx.y * rows_per_group, r_end = min(r_begin + rows_per_group, total_rows);
  for (int r = r_begin + wave; r < r_end; r += 4) {
    const Row q = row_of(r, C, K2);
    const int k = q.kind * 3 + q.stage;                              // loss index: cls 0-2, bbox 3-5, kpt 6-8
    if (q.kind == 0) {
      const long long off = ((long long)b * C + q.c) * N + ic;
      const float x = maps.cls[q.stage][off];
      if (!BACKWARD) {
        const float l = focal_fwd(x, p.label, q.c, cfg.gamma[q.stage], cfg.alpha[q.stage]) * label_w;
        if (live) acc[k] += l;
      } else if (live) {
        // loss = lw * (sum / num_total): d/dx = up * lw / num_total * w * focal'(x)
        const float g = upstream[k] * cfg.loss_weight[k] / num_total_dev[0];
        grads.cls[q.stage][off] = focal_bwd(x, p.label, q.c, cfg.gamma[q.stage], cfg.alpha[q.stage]) * label_w * g;
      }
    } else {
      const float beta = cfg.beta[k - 3];
      float pred_raw, centre, target, w;
      long long off;
      if (q.kind == 1) {            // boxes: channels (x1, y1, x2, y2), offset_to_pts(y_first=False)
        off = ((long long)b * 4 + q.c) * N + ic;
        pred_raw = maps.bbox[q.stage][off];
        centre = (q.c & 1) ? p.py : p.px;
        target = gbox[q.c];
        w = p.a > 0 ? 1.0f : 0.0f;
      } else {                      // keypoints: channel pairs are (y, x); the loss pairs them with (x, y) targets
        off = ((long long)b * K2 + q.c) * N + ic;
        pred_raw = maps.kpt[q.stage][off];
        const int m = q.c >> 1, is_x = q.c & 1;
        centre = is_x ? p.px : p.py;
        target = gkp[m * 3 + (is_x ? 0 : 1)];
        w = (p.a > 0 && gkp[m * 3 + 2] != 0.0f) ? kp_w : 0.0f;
      }
      const float pred = pred_raw * t.stride + centre;
      const float x = pred / nt - target / nt, diff = fabsf(x);      // csrc/smooth_l1.hip's expressions
      if (!BACKWARD) {
        const float l = diff < beta ? 0.5f * diff * diff / beta : diff - 0.5f * beta;
        if (live && w != 0.0f) acc[k] += l * w;
      } else if (live) {
        const float dl = diff < beta ? x / beta : (x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f);
        const float g = upstream[k] * cfg.loss_weight[k] / num_total_dev[0];
        const float gr = w != 0.0f ? g * w * dl / nt * t.stride : 0.0f;
        if (q.kind == 1) grads.bbox[q.stage][off] = gr; else grads.kpt[q.stage][off] = gr;
      }
    }
  }
  if (!BACKWARD) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float v = acc[k];
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_part[wave][k] = v;
    }
    __syncthreads();
    if (tid < 9) {
      const long long wg = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      partial[wg * 9 + tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
    }
  }
  (void)inv_total;
}

// one block: n_pos per image -> num_total; partials in fixed order; the nine losses
__global__ __launch_bounds__(256) void head_loss_finish(const kgdet_head_targets t, const kgdet_head_loss_cfg cfg,
                                                        const float *__restrict__ dsel, int gmax,
                                                        const float *__restrict__ partial, int num_partials,
                                                        float *__restrict__ losses, float *__restrict__ num_total_out) {
  __shared__ float red[4][9];
  __shared__ int s_pos[4];
  __shared__ float s_total;
  const int N = t.H * t.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float total = 0.f;
  for (int b = 0; b < t.B; ++b) {
    int cnt = 0;
    for (int i = tid; i < N; i += 256) cnt += final_assign(dsel, b, gmax, t.num_gt[b], N, i) > 0 ? 1 : 0;
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    __syncthreads();
    if (lane == 0) s_pos[wave] = cnt;
    __syncthreads();
    total += (float)max(s_pos[0] + s_pos[1] + s_pos[2] + s_pos[3], 1);     // point_target_kp.py:60: max(n_pos, 1) per image
  }
  if (tid == 0) { s_total = total; num_total_out[0] = total; }
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  for (int p = tid; p < num_partials; p += 256)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] += partial[(long long)p * 9 + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float v = acc[k];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid < 9) {
    const float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    losses[tid] = cfg.loss_weight[tid] * (s / s_total);                  // losses/utils.py:44-48, focal_loss.py:76-82
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

static int head_check(const kgdet_head_targets *t, const kgdet_head_loss_cfg *cfg, int *gmax) {
  KGDET_CHECK_SHAPE(t && cfg, "null descriptor");
  KGDET_CHECK_SHAPE(t->B >= 1 && t->B <= kMaxImages, "1..%d images per call", kMaxImages);
  KGDET_CHECK_SHAPE(t->H > 0 && t->W > 0 && t->H * t->W <= kMaxPoints, "point grid beyond %d points", kMaxPoints);
  KGDET_CHECK_SHAPE(t->num_classes > 0 && t->num_keypoints > 0, "bad channel counts");
  KGDET_CHECK_SHAPE(cfg->pos_num >= 1 && cfg->normalize_term > 0.0f, "bad assigner / normaliser");
  for (int b = 0; b < t->B; ++b) {
    const int vh = t->valid_h[b] > 0 ? (t->valid_h[b] < t->H ? t->valid_h[b] : t->H) : t->H;
    const int vw = t->valid_w[b] > 0 ? (t->valid_w[b] < t->W ? t->valid_w[b] : t->W) : t->W;
    KGDET_CHECK_SHAPE(t->valid_h[b] >= 0 && t->valid_w[b] >= 0 && (long long)vh * vw >= cfg->pos_num,
                      "image %d: fewer valid points than pos_num", b);
  }
  int g = 0;
  for (int b = 0; b < t->B; ++b) {
    KGDET_CHECK_SHAPE(t->num_gt[b] >= 1 && t->num_gt[b] <= kMaxGt, "image %d: %d ground-truth boxes (1..%d)", b,
                      t->num_gt[b], kMaxGt);
    KGDET_CHECK_SHAPE(t->gt_bboxes[b] && t->gt_keypoints[b], "null ground-truth pointer");
    if (t->num_gt[b] > g) g = t->num_gt[b];
  }
  for (int k = 0; k < 6; ++k) KGDET_CHECK_SHAPE(cfg->beta[k] > 0.0f, "beta must be positive");
  *gmax = g;
  return KGDET_OK;
}

static int head_rows_per_group(const kgdet_head_targets *t, int *groups) {
  const int total_rows = 3 * (t->num_classes + 4 + 2 * t->num_keypoints);
  // enough workgroups for the chip: tiles x groups x images ~ 512
  const int tiles = ceil_div(t->H * t->W, 64);
  int g = ceil_div(512, tiles * t->B);
  if (g < 1) g = 1;
  if (g > total_rows / 4) g = total_rows / 4 > 0 ? total_rows / 4 : 1;
  const int rows = ceil_div(total_rows, g);
  *groups = ceil_div(total_rows, rows);
  return rows;
}

size_t kgdet_head_loss_workspace_bytes(const kgdet_head_targets *t) {
  if (t == nullptr) return 0;
  int groups = 0;
  head_rows_per_group(t, &groups);
  const size_t N = (size_t)t->H * t->W, tiles = (N + 63) / 64;
  return align_up((size_t)t->B * kMaxGt * N * sizeof(float), 256) + align_up(tiles * groups * t->B * 9 * sizeof(float), 256) +
         256;
}

int kgdet_head_loss_forward(const kgdet_head_targets *t, const kgdet_head_loss_cfg *cfg, const kgdet_head_maps *maps,
                            float *losses, float *num_total, void *workspace, size_t workspace_bytes, void *stream) {
  int gmax = 0;
  if (int rc = head_check(t, cfg, &gmax)) return rc;
  KGDET_CHECK_SHAPE(maps && losses && num_total, "null pointer");
  for (int s = 0; s < 3; ++s) KGDET_CHECK_SHAPE(maps->cls[s] && maps->bbox[s] && maps->kpt[s], "null prediction map");
  const size_t need = kgdet_head_loss_workspace_bytes(t);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("head_loss: needs %zu bytes of workspace (kgdet_head_loss_workspace_bytes), got %zu", need, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  const int N = t->H * t->W;
  KGDET_CHECK_SHAPE(cfg->pos_num <= N, "pos_num beyond the number of points");
  float *dsel = reinterpret_cast<float *>(workspace);
  float *partial = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) +
                                             align_up((size_t)t->B * kMaxGt * N * sizeof(float), 256));
  hipLaunchKernelGGL(head_assign_select, dim3(gmax, t->B), dim3(256), (size_t)N * 2 * sizeof(float), (hipStream_t)stream, *t,
                     cfg->pos_num, kMaxGt, dsel);
  KGDET_CHECK_LAUNCH("head_assign_select");
  int groups = 0;
  const int rows = head_rows_per_group(t, &groups);
  const int tiles = ceil_div(N, 64);
  kgdet_head_maps none = {};
  hipLaunchKernelGGL(head_loss_rows<false>, dim3(tiles, groups, t->B), dim3(256), 0, (hipStream_t)stream, *t, *cfg, *maps,
                     dsel, kMaxGt, rows, partial, (const float *)nullptr, (const float *)nullptr, none);
  KGDET_CHECK_LAUNCH("head_loss_rows<forward>");
  hipLaunchKernelGGL(head_loss_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, *t, *cfg, dsel, kMaxGt, partial,
                     tiles * groups * t->B, losses, num_total);
  KGDET_CHECK_LAUNCH("head_loss_finish");
  return KGDET_OK;
}

int kgdet_head_loss_backward(const kgdet_head_targets *t, const kgdet_head_loss_cfg *cfg, const kgdet_head_maps *maps,
                             const float *grad_losses, const float *num_total, const kgdet_head_maps *grads,
                             const void *workspace, size_t workspace_bytes, void *stream) {
  int gmax = 0;
  if (int rc = head_check(t, cfg, &gmax)) return rc;
  KGDET_CHECK_SHAPE(maps && grads && grad_losses && num_total, "null pointer");
  for (int s = 0; s < 3; ++s)
    KGDET_CHECK_SHAPE(maps->cls[s] && maps->bbox[s] && maps->kpt[s] && grads->cls[s] && grads->bbox[s] && grads->kpt[s],
                      "null map");
  KGDET_CHECK_SHAPE(workspace && workspace_bytes >= kgdet_head_loss_workspace_bytes(t),
                    "the forward call's workspace (the per-gt selections) is needed");
  const int N = t->H * t->W;
  const float *dsel = reinterpret_cast<const float *>(workspace);
  int groups = 0;
  const int rows = head_rows_per_group(t, &groups);
  hipLaunchKernelGGL(head_loss_rows<true>, dim3(ceil_div(N, 64), groups, t->B), dim3(256), 0, (hipStream_t)stream, *t, *cfg,
                     *maps, dsel, kMaxGt, rows, (float *)nullptr, num_total, grad_losses, *grads);
  KGDET_CHECK_LAUNCH("head_loss_rows<backward>");
  return KGDET_OK;
}

}  // extern "C"
