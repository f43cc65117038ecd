// fp16 variant (BASELINE configs[3]): the reference's fp32 ONNX export (yoloface/pytorch/yoloface-50k.onnx; the same
// graph as yoloface/pytorch/yoloface.py:83-119) run with fp16 weights and activations, fp32 accumulation.
// Layer by layer over an HBM arena: dense 1x1 convs on v_mfma_f32_16x16x32_f16, depthwise / 3x3 / pools on the VALU.
// This is the tolerance-checked side configuration, not the int8 hot path: correctness and the MFMA-f16 mapping
// matter here, fusion does not (yet).  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "yf_fp16.h"

namespace {

typedef _Float16 half_t;
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.1f * v; }

// conv2d_1: 3x3 stride 2, pad 1, Cin 3 -> Cout 8.  One thread per output pixel (all 8 channels).
__global__ void __launch_bounds__(256) k_conv1(const half_t* __restrict__ in, const half_t* __restrict__ w /*[8][3][3][3]*/,
                                               const float* __restrict__ bias, half_t* __restrict__ out, long n, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * OH * OW) return;
  const long f = i / (OH * OW); const int p = (int)(i - f * OH * OW);
  const int oy = p / OW, ox = p - oy * OW;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = bias[o];
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy + ky - 1;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox + kx - 1;
      if (ix < 0 || ix >= W) continue;
      const half_t* px = in + ((f * H + iy) * W + ix) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = (float)px[c];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] += v * (float)w[((o * 3 + ky) * 3 + kx) * 3 + c];
      }
    }
  }
  half_t* dst = out + ((f * OH + oy) * OW + ox) * 8;
#pragma unroll
  for (int o = 0; o < 8; ++o) dst[o] = (half_t)leaky(acc[o]);
}

// 1x1 conv on MFMA: weights are the A operand (rows = output channels), pixels the B operand (columns), so a lane ends
// up with 4 consecutive channels of one pixel.  Wave = 16 pixels x 16 channels; K in steps of 32 (lane l supplies
// k = 8*(l>>4)+j of its row/column -- the f16 operand map of cdna_hip_programming.md section 3).
// in  : [npix][cs_in] fp16 (cs_in multiple of 8, zero padded), w: [cout_pad16][kpad] fp16 zero padded,
// out : [npix][cs_out] at channel offset ch0; res (optional): [npix][cs_res] added before the activation-less store.
__global__ void __launch_bounds__(256) k_pw_mfma(const half_t* __restrict__ in, int cs_in, const half_t* __restrict__ w, int kpad,
                                                 const float* __restrict__ bias, int cout, half_t* __restrict__ out, int cs_out,
                                                 int ch0, const half_t* __restrict__ res, int cs_res, int act, float* __restrict__ out32,
                                                 long npix) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const long tile = (long)blockIdx.x * 4 + wv;
  const long pix = tile * 16 + c;
  const long pc = pix < npix ? pix : npix - 1;
  const int ntile = (cout + 15) / 16;
  for (int nt = 0; nt < ntile; ++nt) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < kpad; k0 += 32) {
      const v8h a = *reinterpret_cast<const v8h*>(w + (long)(nt * 16 + c) * kpad + k0 + 8 * g);
      v8h b = {0, 0, 0, 0, 0, 0, 0, 0};
      if (k0 + 8 * g < cs_in) b = *reinterpret_cast<const v8h*>(in + pc * cs_in + k0 + 8 * g);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    if (pix < npix) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = nt * 16 + 4 * g + j;
        if (ch < cout) {
          float v = acc[j] + bias[ch];
          if (res) v += (float)res[pix * cs_res + ch];
          if (act) v = leaky(v);
          if (out32) out32[pix * cout + ch] = v;
          else out[pix * cs_out + ch0 + ch] = (half_t)v;
        }
      }
    }
  }
}

// depthwise 3x3 (pad 1), stride 1 or 2, + bias + LeakyReLU.  One thread per (output pixel, channel).
__global__ void __launch_bounds__(256) k_dw3x3(const half_t* __restrict__ in, int cs_in, const half_t* __restrict__ w /*[3][3][c]*/,
                                               const float* __restrict__ bias, half_t* __restrict__ out, int cs_out, long n, int H,
                                               int W, int C, int stride) {
  const int OH = H / stride, OW = W / stride;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * OH * OW * C) return;
  const int ch = (int)(i % C); long t = i / C;
  const int ox = (int)(t % OW); t /= OW;
  const int oy = (int)(t % OH); const long f = t / OH;
  float acc = bias[ch];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = stride * oy + ky - 1;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = stride * ox + kx - 1;
      if (ix < 0 || ix >= W) continue;
      acc += (float)in[((f * H + iy) * W + ix) * cs_in + ch] * (float)w[(ky * 3 + kx) * C + ch];
    }
  }
  out[((f * OH + oy) * OW + ox) * cs_out + ch] = (half_t)leaky(acc);
}

// max-pool k x k, stride 2, pad p (padding never wins); writes at channel offset 0 of a cs_out-strided tensor.
__global__ void __launch_bounds__(256) k_maxpool(const half_t* __restrict__ in, int cs_in, half_t* __restrict__ out, int cs_out,
                                                 long n, int H, int W, int C, int k, int pad) {
  const int OH = H / 2, OW = W / 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * OH * OW * C) return;
  const int ch = (int)(i % C); long t = i / C;
  const int ox = (int)(t % OW); t /= OW;
  const int oy = (int)(t % OH); const long f = t / OH;
  float m = -65504.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = 2 * oy - pad + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = 2 * ox - pad + kx;
      if (ix < 0 || ix >= W) continue;
      m = fmaxf(m, (float)in[((f * H + iy) * W + ix) * cs_in + ch]);
    }
  }
  out[((f * OH + oy) * OW + ox) * cs_out + ch] = (half_t)m;
}

struct ConvW { int dw, cin, cout, k, stride; int kpad, cout_pad; half_t* d_w; float* d_b; };

#define HIPCHK(ctx, call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(rc_); return -1; } } while (0)

}  // namespace

struct yf_fp16 {
  int device = 0;
  std::vector<ConvW> convs;
  half_t* arena = nullptr; long arena_frames = 0;
  std::string err;
};

static int rup(int v, int m) { return (v + m - 1) / m * m; }

extern "C" {

const char* yf_fp16_error(const yf_fp16* c) { return c ? c->err.c_str() : "null context"; }

void yf_fp16_destroy(yf_fp16* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (ConvW& v : c->convs) { if (v.d_w) (void)hipFree(v.d_w); if (v.d_b) (void)hipFree(v.d_b); }
  if (c->arena) (void)hipFree(c->arena);
  delete c;
}

// yfw: the file written by tools/gen_fp16_model.py ('YFW1', 24 convs, fp32 weights OHWI / HWC + bias)
int yf_fp16_create(int device, const void* yfw, size_t bytes, yf_fp16** out, char* err, size_t errlen) {
  auto fail = [&](const std::string& m) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); return -1; };
  if (!yfw || bytes < 8 || !out || memcmp(yfw, "YFW1", 4)) return fail("not a YFW1 weight pack");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  yf_fp16* c = new yf_fp16();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return fail("hipSetDevice failed"); }
  const uint8_t* p = (const uint8_t*)yfw;
  uint32_t n; memcpy(&n, p + 4, 4);
  size_t off = 8;
  for (uint32_t i = 0; i < n; ++i) {
    if (off + 24 > bytes) { yf_fp16_destroy(c); return fail("truncated weight pack"); }
    uint32_t h[6]; memcpy(h, p + off, 24); off += 24;
    ConvW v = {};
    v.dw = (int)h[0]; v.cin = (int)h[1]; v.cout = (int)h[2]; v.k = (int)h[3]; v.stride = (int)h[4];
    const size_t nw = h[5];
    if (off + 4 * (nw + v.cout) > bytes) { yf_fp16_destroy(c); return fail("truncated weight pack"); }
    const float* wf = (const float*)(p + off); off += 4 * nw;
    const float* bf = (const float*)(p + off); off += 4 * (size_t)v.cout;
    std::vector<half_t> wh;
    if (!v.dw && v.k == 1) {                       // [cout_pad16][kpad32], zero padded: the MFMA A operand
      v.kpad = rup(v.cin, 32); v.cout_pad = rup(v.cout, 16);
      wh.assign((size_t)v.cout_pad * v.kpad, (half_t)0);
      for (int o = 0; o < v.cout; ++o) for (int k = 0; k < v.cin; ++k) wh[(size_t)o * v.kpad + k] = (half_t)wf[(size_t)o * v.cin + k];
    } else {
      wh.resize(nw);
      for (size_t k = 0; k < nw; ++k) wh[k] = (half_t)wf[k];
    }
    if (hipMalloc((void**)&v.d_w, wh.size() * sizeof(half_t)) != hipSuccess || hipMalloc((void**)&v.d_b, 4 * (size_t)v.cout) != hipSuccess ||
        hipMemcpy(v.d_w, wh.data(), wh.size() * sizeof(half_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v.d_b, bf, 4 * (size_t)v.cout, hipMemcpyHostToDevice) != hipSuccess) {
      c->convs.push_back(v); yf_fp16_destroy(c); return fail("uploading fp16 weights failed");
    }
    c->convs.push_back(v);
  }
  if (c->convs.size() != 24) { yf_fp16_destroy(c); return fail("expected 24 convolutions"); }
  *out = c;
  return 0;
}

// d_in: fp16 [n][56][56][3] (pixel / 255), d_out: fp32 logits [n][7][7][18]
int yf_fp16_run_device(yf_fp16* c, const void* d_in, void* d_out, long n, void* stream) {
  if (!c || !d_in || !d_out || n < 0) return -2;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  // per-frame arena (halfs): pixel strides are channel counts rounded up to 8
  const long P1 = 784, P2 = 196, P3 = 49;
  const long o_t1 = 0, o_t2 = o_t1 + P1 * 8, o_t3 = o_t2 + P1 * 8, o_t4 = o_t3 + P1 * 8, o_t6 = o_t4 + P1 * 24, o_t7 = o_t6 + P2 * 24,
             o_t8 = o_t7 + P2 * 8, o_t9 = o_t8 + P2 * 40, o_t11 = o_t9 + P2 * 40, o_t14 = o_t11 + P2 * 8, o_t15 = o_t14 + P2 * 40,
             o_t17 = o_t15 + P2 * 24, o_t18 = o_t17 + P3 * 24, o_t19 = o_t18 + P3 * 8, o_t20 = o_t19 + P3 * 40, o_t22 = o_t20 + P3 * 40,
             o_t23 = o_t22 + P3 * 8, o_t24 = o_t23 + P3 * 40, o_t26 = o_t24 + P3 * 40, o_t30 = o_t26 + P3 * 8, o_t31 = o_t30 + P3 * 48,
             o_t32 = o_t31 + P3 * 40, o_t33 = o_t32 + P3 * 40, per_frame_end = o_t33 + P3 * 32;
  // tensors are stored frame-major per buffer: buffer b of all frames = arena + n * o_b
  const long cap = n;
  if (cap > c->arena_frames) {
    if (c->arena) (void)hipFree(c->arena);
    c->arena = nullptr; c->arena_frames = 0;
    HIPCHK(c, hipMalloc((void**)&c->arena, (size_t)cap * per_frame_end * sizeof(half_t)));
    c->arena_frames = cap;
  }
  HIPCHK(c, hipMemsetAsync(c->arena, 0, (size_t)n * per_frame_end * sizeof(half_t), s));   // channel padding must read as 0
  half_t* A = c->arena;
  auto T = [&](long off) { return A + n * off; };
  const half_t* in = (const half_t*)d_in;
  const std::vector<ConvW>& cv = c->convs;
  auto grid = [](long items) { return dim3((unsigned)((items + 255) / 256)); };
  auto pw = [&](int ci, const half_t* x, int cs_in, half_t* y, int cs_out, int ch0, const half_t* res, int cs_res, int act, float* y32, long npix) {
    const ConvW& v = cv[ci];
    hipLaunchKernelGGL(k_pw_mfma, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, s, x, cs_in, v.d_w, v.kpad, v.d_b, v.cout, y, cs_out, ch0,
                       res, cs_res, act, y32, npix);
  };
  auto dw = [&](int ci, const half_t* x, int cs_in, half_t* y, int cs_out, int H, int C) {
    const ConvW& v = cv[ci];
    hipLaunchKernelGGL(k_dw3x3, grid(n * (H / v.stride) * (H / v.stride) * C), dim3(256), 0, s, x, cs_in, v.d_w, v.d_b, y, cs_out, n, H, H, C, v.stride);
  };
  hipLaunchKernelGGL(k_conv1, grid(n * P1), dim3(256), 0, s, in, cv[0].d_w, cv[0].d_b, T(o_t1), n, 56, 56);                // conv2d_1
  dw(1, T(o_t1), 8, T(o_t2), 8, 28, 8);                                                                                       // conv2d_3
  pw(2, T(o_t2), 8, T(o_t3), 8, 0, nullptr, 0, 0, nullptr, n * P1);                                                           // conv2d_5
  pw(3, T(o_t3), 8, T(o_t4), 24, 0, nullptr, 0, 1, nullptr, n * P1);                                                          // conv2d_6
  hipLaunchKernelGGL(k_maxpool, grid(n * P2 * 18), dim3(256), 0, s, T(o_t4), 24, T(o_t14), 40, n, 28, 28, 18, 8, 3);         // pool_8 -> concat[0,18)
  dw(4, T(o_t4), 24, T(o_t6), 24, 28, 18);                                                                                    // conv2d_10
  pw(5, T(o_t6), 24, T(o_t7), 8, 0, nullptr, 0, 0, nullptr, n * P2);                                                          // conv2d_12
  pw(6, T(o_t7), 8, T(o_t8), 40, 0, nullptr, 0, 1, nullptr, n * P2);                                                          // conv2d_13
  dw(7, T(o_t8), 40, T(o_t9), 40, 14, 36);                                                                                    // conv2d_15
  pw(8, T(o_t9), 40, T(o_t11), 8, 0, T(o_t7), 8, 0, nullptr, n * P2);                                                         // conv2d_17 + add
  pw(9, T(o_t11), 8, T(o_t14), 40, 18, nullptr, 0, 1, nullptr, n * P2);                                                       // conv2d_19 -> concat[18,36)
  pw(10, T(o_t14), 40, T(o_t15), 24, 0, nullptr, 0, 1, nullptr, n * P2);                                                      // conv2d_23
  hipLaunchKernelGGL(k_maxpool, grid(n * P3 * 24), dim3(256), 0, s, T(o_t15), 24, T(o_t30), 48, n, 14, 14, 24, 4, 1);        // pool_25 -> concat[0,24)
  dw(11, T(o_t15), 24, T(o_t17), 24, 14, 24);                                                                                 // conv2d_27
  pw(12, T(o_t17), 24, T(o_t18), 8, 0, nullptr, 0, 0, nullptr, n * P3);                                                       // conv2d_29
  pw(13, T(o_t18), 8, T(o_t19), 40, 0, nullptr, 0, 1, nullptr, n * P3);                                                       // conv2d_30
  dw(14, T(o_t19), 40, T(o_t20), 40, 7, 40);                                                                                  // conv2d_32
  pw(15, T(o_t20), 40, T(o_t22), 8, 0, T(o_t18), 8, 0, nullptr, n * P3);                                                      // conv2d_34 + add
  pw(16, T(o_t22), 8, T(o_t23), 40, 0, nullptr, 0, 1, nullptr, n * P3);                                                       // conv2d_36
  dw(17, T(o_t23), 40, T(o_t24), 40, 7, 40);                                                                                  // conv2d_38
  pw(18, T(o_t24), 40, T(o_t26), 8, 0, T(o_t22), 8, 0, nullptr, n * P3);                                                      // conv2d_40 + add
  pw(19, T(o_t26), 8, T(o_t30), 48, 24, nullptr, 0, 1, nullptr, n * P3);                                                      // conv2d_42 -> concat[24,48)
  pw(20, T(o_t30), 48, T(o_t31), 40, 0, nullptr, 0, 1, nullptr, n * P3);                                                      // conv2d_47
  dw(21, T(o_t31), 40, T(o_t32), 40, 7, 40);                                                                                  // conv2d_49
  pw(22, T(o_t32), 40, T(o_t33), 32, 0, nullptr, 0, 1, nullptr, n * P3);                                                      // conv2d_51
  pw(23, T(o_t33), 32, nullptr, 0, 0, nullptr, 0, 0, (float*)d_out, n * P3);                                                  // head (fp32 logits)
  HIPCHK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
