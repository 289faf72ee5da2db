// Softmax cross-entropy losses of the PixelLink-style heads, with online hard negative
// mining (OHNM) and the focal variant.
//
//   pixel_rule 0  nets/model.py:204-261       OHNM per image: k = min(3*n_pos, n_neg) hardest
//                                             negatives (tf.nn.top_k on -P(neg), tie-inclusive);
//                                             pixel = sum(CE*sel)/n_pos_batch, 0 if no positives
//   pixel_rule 1  nets/model_vgg_16.py:243-282 `ohem_loss`: weights = positives only
//   pixel_rule 2  nets/pixellink.py:88-263     `PixelLinkNet.build_loss`: mean CE over all pixels
//   link_gate 1   link weights W_pos/W_neg multiplied by the selected pixel mask, NO zero guard
//                 (model.py:238-254 -> NaN when a direction has no positive or no negative link)
//   link_gate 0   ungated, zero-guarded (pixellink.py:198-212)
//   focal 1       link CE replaced by FL = -alpha_t (1-p_t)^gamma log p_t (Lin et al. 2017;
//                 absent from the reference tree, SURVEY D1: build-defined, parity unpinned)
//
// total = sum_i link_i + 2 * pixel.  The mined mask is a constant w.r.t. the gradient.
//
// OHNM threshold: exact k-th smallest P(neg) among an image's negatives by a 4-pass 8-bit radix
// select on the float bits (one workgroup per image, integer LDS histograms: deterministic),
// then `score <= threshold` — the same float expression in every pass.
#include "common.h"

namespace {

struct SlP {
  int n, hw, pixel_rule, label_rule, link_gate, focal;
  float neg_ratio, alpha, gamma;
};

__device__ __forceinline__ bool is_pos(float label, int rule) { return rule ? label > 0.f : label == 1.f; }
__device__ __forceinline__ bool is_neg(float label, int rule) { return rule ? !(label > 0.f) : label == 0.f; }

// exp(x) as a FIXED sequence of IEEE f32 operations (no FMA contraction, no library call): the
// CPU checker evaluates the same sequence in numpy float32, so the mining scores — and with them the
// k-th-smallest threshold and the mined mask, which are index work — agree bit for bit.
// Range reduction x = n ln2 + r (Cody-Waite), degree-6 Taylor in r (|r| <= 0.347: ~1 ulp), 2^n scale.
__device__ __forceinline__ float det_exp(float x) {
#pragma clang fp contract(off)
  x = fminf(fmaxf(x, -80.f), 80.f);
  const float n = rintf(x * 1.44269504f);
  float r = x - n * 0.693145751953125f;
  r = r - n * 1.42860677e-06f;
  float p = 1.3888889e-03f;
  p = p * r + 8.3333338e-03f;
  p = p * r + 4.1666668e-02f;
  p = p * r + 1.6666667e-01f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  return ldexpf(p, (int)n);
}

// P(class 0) of a 2-way softmax, = exp(l0) / (exp(l0) + exp(l1))
__device__ __forceinline__ float neg_score(float l0, float l1) {
#pragma clang fp contract(off)
  return 1.f / (1.f + det_exp(l1 - l0));
}

// 2-class CE and softmax of logits (l0, l1) for class t
__device__ __forceinline__ float ce2(float l0, float l1, int t, float* p1) {
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float s = e0 + e1;
  *p1 = e1 / s;
  return logf(s) + m - (t ? l1 : l0);
}

__device__ __forceinline__ float focal2(float l0, float l1, int t, float alpha, float gamma, float* dz1) {
  // L = -a_t (1-p)^g log p,  p = softmax prob of the true class; dz1 = dL/d(l1)
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float s = e0 + e1;
  const float p = (t ? e1 : e0) / s;
  const float logp = (t ? l1 : l0) - m - logf(s);
  const float a = t ? alpha : 1.f - alpha;
  const float om = 1.f - p;
  const float omg = powf(fmaxf(om, 1e-30f), gamma);
  const float dzt = a * (gamma * omg * p * logp - omg * om);   // dL/dz_true
  *dz1 = t ? dzt : -dzt;
  return -a * omg * logp;
}

// ------------------------------------------------------------- OHNM threshold per image
__global__ __launch_bounds__(1024) void ohnm_threshold_kernel(SlP p, const float* __restrict__ pl,
                                                              const float* __restrict__ lab,
                                                              float* __restrict__ thr) {
  __shared__ int hist[256];
  __shared__ int s_cnt[2];
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  const int img = blockIdx.x;
  const float* l = pl + (size_t)img * p.hw * 2;
  const float* y = lab + (size_t)img * p.hw;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int np = 0, nn = 0;
  for (int i = threadIdx.x; i < p.hw; i += 1024) {
    np += is_pos(y[i], p.label_rule) ? 1 : 0;
    nn += is_neg(y[i], p.label_rule) ? 1 : 0;
  }
  atomicAdd(&s_cnt[0], np);
  atomicAdd(&s_cnt[1], nn);
  __syncthreads();
  const int n_pos = s_cnt[0], n_neg = s_cnt[1];
  int k = (int)fminf((float)n_pos * p.neg_ratio, (float)n_neg);
  if (n_pos == 0 || k <= 0) {
    if (threadIdx.x == 0) thr[img] = -1.f;   // nothing is selected (scores are >= 0)
    return;
  }
  if (threadIdx.x == 0) { s_prefix = 0u; s_k = k; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = threadIdx.x; i < p.hw; i += 1024) {
      if (!is_neg(y[i], p.label_rule)) continue;
      const unsigned u = __float_as_uint(neg_score(l[2 * i], l[2 * i + 1]));
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int kk = s_k, cum = 0, d = 0;
      for (; d < 256; ++d) {
        if (cum + hist[d] >= kk) break;
        cum += hist[d];
      }
      s_prefix = prefix | ((unsigned)d << shift);
      s_k = kk - cum;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) thr[img] = __uint_as_float(s_prefix);
}

// per-pixel weight of the pixel CE term
__device__ __forceinline__ float pixel_weight(const SlP& p, float label, float l0, float l1, float thr) {
  if (p.pixel_rule == 2) return 1.f;
  if (is_pos(label, p.label_rule)) return 1.f;
  if (p.pixel_rule == 0 && is_neg(label, p.label_rule) && neg_score(l0, l1) <= thr) return 1.f;
  return 0.f;
}

// sums: [0] sum CE*W  [1] n_pos (batch)  then per direction i: [2+4i] sum L*Wp [3+4i] sum Wp
// [4+4i] sum L*Wn [5+4i] sum Wn
constexpr int NS = 34;

__global__ __launch_bounds__(256) void sl_reduce_kernel(SlP p, const float* __restrict__ pl,
                                                        const float* __restrict__ ll,
                                                        const float* __restrict__ plab,
                                                        const float* __restrict__ llab,
                                                        const float* __restrict__ thr,
                                                        float* __restrict__ partial) {
  __shared__ float red[4][NS];
  float s[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) s[j] = 0.f;
  const size_t total = (size_t)p.n * p.hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / p.hw);
    const float y = plab[i], l0 = pl[2 * i], l1 = pl[2 * i + 1];
    const bool pos = is_pos(y, p.label_rule);
    const float W = pixel_weight(p, y, l0, l1, p.pixel_rule == 0 ? thr[img] : 0.f);
    float p1;
    const float ce = ce2(l0, l1, pos ? 1 : 0, &p1);
    s[0] += ce * W;
    s[1] += pos ? 1.f : 0.f;
    const float G = p.link_gate ? W : 1.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float yl = llab[i * 8 + d];
      const float a0 = ll[i * 16 + 2 * d], a1 = ll[i * 16 + 2 * d + 1];
      const bool lp = is_pos(yl, p.label_rule), ln = is_neg(yl, p.label_rule);
      float dummy;
      const float L = p.focal ? focal2(a0, a1, lp ? 1 : 0, p.alpha, p.gamma, &dummy)
                              : ce2(a0, a1, lp ? 1 : 0, &dummy);
      const float wp = lp ? G : 0.f, wn = ln ? G : 0.f;
      s[2 + 4 * d] += L * wp;
      s[3 + 4 * d] += wp;
      s[4 + 4 * d] += L * wn;
      s[5 + 4 * d] += wn;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const float v = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS)
    partial[(size_t)blockIdx.x * NS + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// sums[34]; loss[0] total, loss[1] pixel term (before the factor 2), loss[2..9] link terms
__global__ void sl_finalize_kernel(SlP p, const float* __restrict__ partial, int T,
                                   float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double part[4][64];
  __shared__ float tot[NS];
  const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
  double a = 0.0;
  if (j < NS)
    for (int t = g; t < T; t += 4) a += (double)partial[(size_t)t * NS + j];
  part[g][j] = a;
  __syncthreads();
  if (threadIdx.x < NS) {
    const float v = (float)(part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
    tot[threadIdx.x] = v;
    sums[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float pixel;
    if (p.pixel_rule == 2) pixel = tot[0] / ((float)p.n * (float)p.hw);
    else if (p.pixel_rule == 0) pixel = tot[1] > 0.f ? tot[0] / tot[1] : 0.f;
    else pixel = tot[0] / tot[1];                      // ohem_loss: no guard in the reference
    float total = 2.f * pixel;
    loss[1] = pixel;
    for (int d = 0; d < 8; ++d) {
      const float lp = tot[2 + 4 * d], wp = tot[3 + 4 * d], ln = tot[4 + 4 * d], wn = tot[5 + 4 * d];
      float v;
      if (p.link_gate) v = lp / wp + ln / wn;           // model.py:249-254: unguarded
      else v = (wp == 0.f ? 0.f : lp / wp) + (wn == 0.f ? 0.f : ln / wn);
      loss[2 + d] = v;
      total += v;
    }
    loss[0] = total;
  }
}

__global__ __launch_bounds__(256) void sl_bwd_kernel(SlP p, const float* __restrict__ pl,
                                                     const float* __restrict__ ll,
                                                     const float* __restrict__ plab,
                                                     const float* __restrict__ llab,
                                                     const float* __restrict__ thr,
                                                     const float* __restrict__ sums, float gscale,
                                                     float* __restrict__ dpl, float* __restrict__ dll) {
  __shared__ float c_pix, c_pos[8], c_neg[8];
  if (threadIdx.x == 0) {
    if (p.pixel_rule == 2) c_pix = 2.f * gscale / ((float)p.n * (float)p.hw);
    else if (p.pixel_rule == 0) c_pix = sums[1] > 0.f ? 2.f * gscale / sums[1] : 0.f;
    else c_pix = 2.f * gscale / sums[1];
  }
  if (threadIdx.x < 8) {
    const float wp = sums[3 + 4 * threadIdx.x], wn = sums[5 + 4 * threadIdx.x];
    c_pos[threadIdx.x] = (!p.link_gate && wp == 0.f) ? 0.f : gscale / wp;
    c_neg[threadIdx.x] = (!p.link_gate && wn == 0.f) ? 0.f : gscale / wn;
  }
  __syncthreads();
  const size_t total = (size_t)p.n * p.hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / p.hw);
    const float y = plab[i], l0 = pl[2 * i], l1 = pl[2 * i + 1];
    const bool pos = is_pos(y, p.label_rule);
    const float W = pixel_weight(p, y, l0, l1, p.pixel_rule == 0 ? thr[img] : 0.f);
    float p1;
    ce2(l0, l1, pos ? 1 : 0, &p1);
    const float g1 = (p1 - (pos ? 1.f : 0.f)) * W * c_pix;   // d/dl1 ; d/dl0 = -g1
    dpl[2 * i] = -g1;
    dpl[2 * i + 1] = g1;
    const float G = p.link_gate ? W : 1.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float yl = llab[i * 8 + d];
      const float a0 = ll[i * 16 + 2 * d], a1 = ll[i * 16 + 2 * d + 1];
      const bool lp = is_pos(yl, p.label_rule), ln = is_neg(yl, p.label_rule);
      float dz1;
      if (p.focal) {
        focal2(a0, a1, lp ? 1 : 0, p.alpha, p.gamma, &dz1);
      } else {
        float q1;
        ce2(a0, a1, lp ? 1 : 0, &q1);
        dz1 = q1 - (lp ? 1.f : 0.f);
      }
      // weights that are exactly zero contribute nothing even when the normaliser is 0/0
      float coef = 0.f;
      if (lp && G != 0.f) coef += G * c_pos[d];
      if (ln && G != 0.f) coef += G * c_neg[d];
      const float gl = dz1 * coef;
      dll[i * 16 + 2 * d] = -gl;
      dll[i * 16 + 2 * d + 1] = gl;
    }
  }
}

// the pixel-term weight map W as bytes (1 = positive or mined negative): what `OHNM_batch` returns
// (nets/model.py:186-197), straight from the expression the loss kernels use
__global__ void sl_selected_kernel(SlP p, const float* __restrict__ pl, const float* __restrict__ plab,
                                   const float* __restrict__ thr, unsigned char* __restrict__ mask) {
  const size_t total = (size_t)p.n * p.hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / p.hw);
    mask[i] = pixel_weight(p, plab[i], pl[2 * i], pl[2 * i + 1], p.pixel_rule == 0 ? thr[img] : 0.f) != 0.f;
  }
}

// ---- the reference's mining helpers as entry points of their own (nets/model.py:161-201) ----------------------------
// get_pos_and_neg_masks: pos = (labels == 1), neg = (labels == 0) (label_rule 0) as bytes
__global__ void label_masks_kernel(const float* __restrict__ lab, size_t total, int rule, unsigned char* __restrict__ pos,
                                   unsigned char* __restrict__ neg) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float y = lab[i];
    pos[i] = is_pos(y, rule) ? 1 : 0;
    neg[i] = is_neg(y, rule) ? 1 : 0;
  }
}

// OHNM_single_image / OHNM_batch on GIVEN scores and masks: per image k = min(n_pos * ratio, #neg), threshold = the k-th
// smallest score among the negatives (exact: 4-pass 8-bit radix select on an order-preserving key of the float bits —
// sign bit flipped for non-negative values, all bits for negative ones — so ANY finite scores work, logits included:
// the reference's top_k(-neg_conf) has no sign restriction, nets/model.py:161-184), selected negatives = neg & score <=
// threshold (tie-inclusive, like `tf.nn.top_k` + `<=`; -0.0 and +0.0 are one value there as here); nothing when
// n_pos == 0.  k = min(n_pos * ratio, #neg) in double (exact for every count a map can hold).
__global__ __launch_bounds__(1024) void ohnm_select_kernel(int hw, float ratio, const float* __restrict__ scores,
                                                           const unsigned char* __restrict__ pos_all,
                                                           const unsigned char* __restrict__ neg_all,
                                                           const int* __restrict__ n_pos_in, float* __restrict__ sel_neg,
                                                           float* __restrict__ selected) {
  __shared__ int hist[256];
  __shared__ int s_cnt[2];
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  const int img = blockIdx.x;
  const float* sc = scores + (size_t)img * hw;
  const unsigned char* pos = pos_all ? pos_all + (size_t)img * hw : nullptr;
  const unsigned char* neg = neg_all + (size_t)img * hw;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int np = 0, nn = 0;
  for (int i = threadIdx.x; i < hw; i += 1024) {
    np += (pos && pos[i]) ? 1 : 0;
    nn += neg[i] ? 1 : 0;
  }
  atomicAdd(&s_cnt[0], np);
  atomicAdd(&s_cnt[1], nn);
  __syncthreads();
  const int n_pos = n_pos_in ? n_pos_in[img] : s_cnt[0], n_neg = s_cnt[1];
  const int k = (int)fmin((double)n_pos * (double)ratio, (double)n_neg);
  float thr = 0.f;
  const bool any = n_pos > 0 && k > 0;                // otherwise nothing is selected
  if (any) {
    if (threadIdx.x == 0) { s_prefix = 0u; s_k = k; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (threadIdx.x < 256) hist[threadIdx.x] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix;
      const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = threadIdx.x; i < hw; i += 1024) {
        if (!neg[i]) continue;
        const unsigned b = __float_as_uint(sc[i]);
        const unsigned u = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);       // unsigned order of u = float order of the score
        if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int kk = s_k, cum = 0, d = 0;
        for (; d < 256; ++d) {
          if (cum + hist[d] >= kk) break;
          cum += hist[d];
        }
        s_prefix = prefix | ((unsigned)d << shift);
        s_k = kk - cum;
      }
      __syncthreads();
    }
    const unsigned key = s_prefix;
    thr = __uint_as_float(key ^ ((key >> 31) ? 0x80000000u : 0xffffffffu));
  }
  for (int i = threadIdx.x; i < hw; i += 1024) {
    const float sn = (any && neg[i] && sc[i] <= thr) ? 1.f : 0.f;
    if (sel_neg) sel_neg[(size_t)img * hw + i] = sn;
    if (selected) selected[(size_t)img * hw + i] = ((pos && pos[i]) ? 1.f : 0.f) + sn;   // cast(pos_mask) + selected_neg_mask
  }
}

// cal_link_loss (nets/model_vgg_16.py:227-241) for ONE direction: CE of the logit pairs against link_gt, weighted by
// W_pixel, sum(CE * Wpos) / sum(Wpos) + sum(CE * Wneg) / sum(Wneg), unguarded like the reference.  Rows may be strided
// (a tf.split slice of the 8- / 16-channel maps): element r of gt at gt[r * gs], pair r at pred[r * ps .. + 1].
__global__ __launch_bounds__(256) void link_ce_reduce_kernel(const float* __restrict__ gt, int gs, const float* __restrict__ pred,
                                                             int ps, const float* __restrict__ W, size_t P,
                                                             float* __restrict__ partial) {
  __shared__ float red[4][4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (size_t)gridDim.x * 256) {
    const float y = gt[i * gs], w = W[i];
    const bool lp = is_pos(y, 0), ln = is_neg(y, 0);
    float dummy;
    const float L = ce2(pred[i * ps], pred[i * ps + 1], lp ? 1 : 0, &dummy);
    const float wp = lp ? w : 0.f, wn = ln ? w : 0.f;
    s[0] += L * wp; s[1] += wp; s[2] += L * wn; s[3] += wn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void link_ce_finalize_kernel(const float* __restrict__ partial, int T, float* __restrict__ sums4, float* __restrict__ loss) {
  __shared__ double part[64][4];
  const int j = threadIdx.x & 3, g = threadIdx.x >> 2;
  double a = 0.0;
  for (int t = g; t < T; t += 64) a += (double)partial[(size_t)t * 4 + j];
  part[g][j] = a;
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int k = 0; k < 64; ++k) v += part[k][threadIdx.x];
    sums4[threadIdx.x] = (float)v;
  }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = sums4[0] / sums4[1] + sums4[2] / sums4[3];
}
__global__ __launch_bounds__(256) void link_ce_bwd_kernel(const float* __restrict__ gt, int gs, const float* __restrict__ pred,
                                                          int ps, const float* __restrict__ W, size_t P,
                                                          const float* __restrict__ sums4, float gscale,
                                                          float* __restrict__ dpred, int ds) {
  const float cp = gscale / sums4[1], cn = gscale / sums4[3];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (size_t)gridDim.x * 256) {
    const float y = gt[i * gs], w = W[i];
    const bool lp = is_pos(y, 0), ln = is_neg(y, 0);
    float q1;
    ce2(pred[i * ps], pred[i * ps + 1], lp ? 1 : 0, &q1);
    float coef = 0.f;                                   // weights that are exactly zero contribute nothing even at 0/0
    if (lp && w != 0.f) coef += w * cp;
    if (ln && w != 0.f) coef += w * cn;
    const float gl = (q1 - (lp ? 1.f : 0.f)) * coef;
    dpred[i * ds] = -gl;
    dpred[i * ds + 1] = gl;
  }
}

int sl_blocks(size_t total) {
  size_t b = (total + 256 * 4 - 1) / (256 * 4);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

int fill(const ocr_softmax_loss_desc* d, SlP* p) {
  OCR_CHECK_ARG(d != nullptr && d->n > 0 && d->hw > 0);
  OCR_CHECK_ARG(d->pixel_rule >= 0 && d->pixel_rule <= 2);
  p->n = d->n; p->hw = d->hw; p->pixel_rule = d->pixel_rule; p->label_rule = d->label_rule;
  p->link_gate = d->link_gate; p->focal = d->focal;
  p->neg_ratio = d->neg_ratio; p->alpha = d->alpha; p->gamma = d->gamma;
  return OCR_OK;
}

}  // namespace

extern "C" size_t ocr_softmax_loss_workspace(const ocr_softmax_loss_desc* d) {
  if (!d) return 0;
  return (size_t)sl_blocks((size_t)d->n * d->hw) * NS * sizeof(float);
}

extern "C" int ocr_softmax_loss_fwd(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                                    const void* link_logits, const void* pixel_labels,
                                    const void* link_labels, void* ohnm_threshold, void* sums34,
                                    void* loss10, void* workspace, size_t ws_bytes, void* stream) {
  SlP p;
  int rc = fill(d, &p);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(pixel_logits && link_logits && pixel_labels && link_labels && ohnm_threshold);
  OCR_CHECK_ARG(sums34 && loss10 && workspace);
  if (ws_bytes < ocr_softmax_loss_workspace(d)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* pl = static_cast<const float*>(pixel_logits);
  const float* plab = static_cast<const float*>(pixel_labels);
  if (p.pixel_rule == 0)
    hipLaunchKernelGGL(ohnm_threshold_kernel, dim3(p.n), dim3(1024), 0, st, p, pl, plab,
                       static_cast<float*>(ohnm_threshold));
  const int T = sl_blocks((size_t)p.n * p.hw);
  hipLaunchKernelGGL(sl_reduce_kernel, dim3(T), dim3(256), 0, st, p, pl,
                     static_cast<const float*>(link_logits), plab,
                     static_cast<const float*>(link_labels),
                     static_cast<const float*>(ohnm_threshold), static_cast<float*>(workspace));
  hipLaunchKernelGGL(sl_finalize_kernel, dim3(1), dim3(256), 0, st, p,
                     static_cast<const float*>(workspace), T, static_cast<float*>(sums34),
                     static_cast<float*>(loss10));
  return ocr_launch_status();
}

extern "C" int ocr_softmax_loss_selected(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                                         const void* pixel_labels, const void* ohnm_threshold,
                                         void* mask_u8, void* stream) {
  SlP p;
  int rc = fill(d, &p);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(pixel_logits && pixel_labels && ohnm_threshold && mask_u8);
  hipLaunchKernelGGL(sl_selected_kernel, dim3(sl_blocks((size_t)p.n * p.hw)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const float*>(pixel_logits),
                     static_cast<const float*>(pixel_labels), static_cast<const float*>(ohnm_threshold),
                     static_cast<unsigned char*>(mask_u8));
  return ocr_launch_status();
}

extern "C" int ocr_softmax_loss_bwd(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                                    const void* link_logits, const void* pixel_labels,
                                    const void* link_labels, const void* ohnm_threshold,
                                    const void* sums34, float grad_scale, void* d_pixel_logits,
                                    void* d_link_logits, void* stream) {
  SlP p;
  int rc = fill(d, &p);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(pixel_logits && link_logits && pixel_labels && link_labels && ohnm_threshold);
  OCR_CHECK_ARG(sums34 && d_pixel_logits && d_link_logits);
  const int T = sl_blocks((size_t)p.n * p.hw) * 2;
  hipLaunchKernelGGL(sl_bwd_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                     static_cast<const float*>(pixel_logits), static_cast<const float*>(link_logits),
                     static_cast<const float*>(pixel_labels), static_cast<const float*>(link_labels),
                     static_cast<const float*>(ohnm_threshold), static_cast<const float*>(sums34),
                     grad_scale, static_cast<float*>(d_pixel_logits),
                     static_cast<float*>(d_link_logits));
  return ocr_launch_status();
}

extern "C" int ocr_label_masks(const void* labels, int64_t count, int label_rule, void* pos_u8, void* neg_u8, void* stream) {
  OCR_CHECK_ARG(labels && pos_u8 && neg_u8 && count > 0 && (label_rule == 0 || label_rule == 1));
  hipLaunchKernelGGL(label_masks_kernel, dim3(sl_blocks((size_t)count)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(labels), (size_t)count, label_rule, static_cast<unsigned char*>(pos_u8),
                     static_cast<unsigned char*>(neg_u8));
  return ocr_launch_status();
}

extern "C" int ocr_ohnm_select(const void* scores, const void* pos_mask_u8, const void* neg_mask_u8, const void* n_pos_i32,
                               int n, int hw, float neg_ratio, void* selected_neg_f32, void* selected_f32, void* stream) {
  OCR_CHECK_ARG(scores && neg_mask_u8 && (pos_mask_u8 || n_pos_i32) && (selected_neg_f32 || selected_f32));
  OCR_CHECK_ARG(n > 0 && hw > 0 && neg_ratio >= 0.f);
  hipLaunchKernelGGL(ohnm_select_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), hw, neg_ratio,
                     static_cast<const float*>(scores), static_cast<const unsigned char*>(pos_mask_u8),
                     static_cast<const unsigned char*>(neg_mask_u8), static_cast<const int*>(n_pos_i32),
                     static_cast<float*>(selected_neg_f32), static_cast<float*>(selected_f32));
  return ocr_launch_status();
}

extern "C" size_t ocr_link_ce_workspace(int64_t count) { return (size_t)sl_blocks((size_t)(count > 0 ? count : 1)) * 4 * sizeof(float); }

extern "C" int ocr_link_ce_fwd(const void* link_gt, int gt_stride, const void* link_pred, int pred_stride, const void* w_pixel,
                               int64_t count, void* sums4, void* loss1, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(link_gt && link_pred && w_pixel && sums4 && loss1 && workspace && count > 0 && gt_stride >= 1 && pred_stride >= 2);
  if (ws_bytes < ocr_link_ce_workspace(count)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = sl_blocks((size_t)count);
  hipLaunchKernelGGL(link_ce_reduce_kernel, dim3(T), dim3(256), 0, st, static_cast<const float*>(link_gt), gt_stride,
                     static_cast<const float*>(link_pred), pred_stride, static_cast<const float*>(w_pixel), (size_t)count,
                     static_cast<float*>(workspace));
  hipLaunchKernelGGL(link_ce_finalize_kernel, dim3(1), dim3(256), 0, st, static_cast<const float*>(workspace), T,
                     static_cast<float*>(sums4), static_cast<float*>(loss1));
  return ocr_launch_status();
}

extern "C" int ocr_link_ce_bwd(const void* link_gt, int gt_stride, const void* link_pred, int pred_stride, const void* w_pixel,
                               int64_t count, const void* sums4, float grad_scale, void* d_link_pred, int d_stride,
                               void* stream) {
  OCR_CHECK_ARG(link_gt && link_pred && w_pixel && sums4 && d_link_pred && count > 0 && gt_stride >= 1 && pred_stride >= 2 &&
                d_stride >= 2);
  hipLaunchKernelGGL(link_ce_bwd_kernel, dim3(sl_blocks((size_t)count) * 2), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(link_gt), gt_stride, static_cast<const float*>(link_pred), pred_stride,
                     static_cast<const float*>(w_pixel), (size_t)count, static_cast<const float*>(sums4), grad_scale,
                     static_cast<float*>(d_link_pred), d_stride);
  return ocr_launch_status();
}
