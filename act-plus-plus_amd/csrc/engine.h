// Internal context of libactmi (not part of the C ABI).
#pragma once
#include "common.h"

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

struct Param {
    std::string key;
    std::vector<int64_t> shape;
    int64_t numel;
    int64_t off;       // float offset into the parameter arena
    bool is_buffer;
};

struct ConvLayer {
    std::string name, bn;
    int cin, cout, k, stride, pad, H, W, Ho, Wo;
    int K = 0;
    float* w = nullptr;       // [cam][cout][K], K index (r,s,c)
    float* scale = nullptr;   // [cam][cout]
    float* bias = nullptr;
};

struct MhaW { float *in_w, *in_b, *out_w, *out_b; };
struct EncW { MhaW attn; float *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b; };
struct DecW { MhaW self_attn, cross; float *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b, *n3w, *n3b; };

struct DbgView { const float* ptr; int64_t numel; };

struct actmi_ctx {
    actmi_config cfg;
    std::string err;
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    std::vector<void*> allocs;
    float* pbase = nullptr;
    int64_t ptotal = 0;
    bool finalized = false;
    // geometry
    int H1, W1, H2, W2, fh, fw, P_, N;
    // prepared weights
    std::vector<ConvLayer> convs;
    float *conv1_w = nullptr, *conv1_scale = nullptr, *conv1_bias = nullptr, *lut = nullptr;
    float *pos_tokens = nullptr, *dec_t1 = nullptr, *dec_q = nullptr, *tmp_vec = nullptr;
    int* rowmap = nullptr;
    int rowmap_B = -1;
    std::vector<EncW> enc, cvae;
    std::vector<DecW> dec;
    // activations
    float *act1 = nullptr, *buf[3] = {nullptr, nullptr, nullptr};
    float *X = nullptr, *X1 = nullptr, *Y = nullptr, *ATT = nullptr, *QKV = nullptr, *Hb = nullptr;
    float *dO = nullptr, *dY = nullptr, *dT2 = nullptr, *dH = nullptr, *hs = nullptr;
    std::map<std::string, DbgView> dbg;
    std::string stop_stage;   // debug: return from the forward right after this stage

    float* P(const std::string& key);
};

int engine_create(const actmi_config* cfg, actmi_ctx** out);
int engine_destroy(actmi_ctx* ctx);
const char* engine_create_error();
int engine_finalize(actmi_ctx* ctx, hipStream_t st);
int engine_forward_infer(actmi_ctx* ctx, const float* qpos, const void* image, int fmt, int B, float* a_hat,
                         hipStream_t st);
