// Host-side helpers shared by the inference translation units: state_dict blob walking and
// eval-mode BatchNorm folding (reference layers: src/alphazero_simple/resnet.py:13-27,44-72).
#pragma once
#include <math.h>
#include <stddef.h>

#include <vector>

namespace azsp {

constexpr int POL_CH = 32;      // policy head conv channels (resnet.py:57)
constexpr int VAL_CH = 3;       // value head conv channels  (resnet.py:66)
constexpr float BN_EPS = 1e-5f;

struct ConvBN { const float *w, *b, *gamma, *beta, *mean, *var; int oc, ic, k; };

static inline const float *take(const float *&p, size_t n) { const float *r = p; p += n; return r; }
static inline ConvBN take_conv_bn(const float *&p, int oc, int ic, int k) {
    ConvBN c;
    c.oc = oc; c.ic = ic; c.k = k;
    c.w = take(p, (size_t)oc * ic * k * k);
    c.b = take(p, oc);
    c.gamma = take(p, oc); c.beta = take(p, oc); c.mean = take(p, oc); c.var = take(p, oc);
    return c;
}
// eval-mode BatchNorm folded into the convolution: y = (conv(x)+b-mean)*gamma/sqrt(var+eps)+beta
static inline void fold(const ConvBN &c, std::vector<float> &w, std::vector<float> &b) {
    const size_t per = (size_t)c.ic * c.k * c.k;
    w.resize((size_t)c.oc * per);
    b.resize(c.oc);
    for (int o = 0; o < c.oc; ++o) {
        const double s = (double)c.gamma[o] / sqrt((double)c.var[o] + (double)BN_EPS);
        for (size_t i = 0; i < per; ++i) w[o * per + i] = (float)((double)c.w[o * per + i] * s);
        b[o] = (float)(((double)c.b[o] - (double)c.mean[o]) * s + (double)c.beta[o]);
    }
}


}  // namespace azsp
