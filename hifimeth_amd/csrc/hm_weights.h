// hm_weights.h -- model files -> canonical fp32 parameters -> MFMA-fragment-packed device blob.
#pragma once
#include <string>
#include <vector>

#include "hm_device.h"

namespace hm {

// Canonical parameters of one context model (BatchNorms after the convs are already folded in the
// shipped ONNX files; bn0 is explicit).  Same tensor order as the .hmw container
// (hifimeth_amd/onnx_weights.py).
struct HostModel {
    int k1 = 0;
    float bn_eps = 1e-5f;
    float bn_gamma[8], bn_beta[8], bn_mean[8], bn_var[8];
    std::vector<float> conv_w[8];  // [Cout][Cin][k]  (ONNX OIW)
    std::vector<float> conv_b[8];
    int kernel[8];
    std::vector<float> fc1_w, fc1_b, fc2_w, fc2_b;  // [256][128], [256], [2][256], [2]
};

struct PackedModel {
    std::vector<float> blob;
    size_t wfrag_off[9], bias_off[9], fc2_w_off, fc2_b_off, bn_off;
    size_t wfrag_h_off[9], bn_h_off;  // fp16 hi/lo data, offsets in floats like the others
    size_t c1f_off, c1f_bias_off, c1f_corr_off;  // conv1 with bn0's one-hot half folded into the weights (see pack_model)
};

// <dir>/<name>.hmw, else <dir>/<name>.onnx (the reference's model_dir layout, mod_main.cpp:76,85,94)
bool load_model_dir(const char* dir, const char* name, HostModel& out, std::string& err);
bool load_hmw(const std::string& path, HostModel& out, std::string& err);
bool load_onnx(const std::string& path, HostModel& out, std::string& err);
PackedModel pack_model(const HostModel& m);
bool save_hmw(const HostModel& m, const std::string& path, std::string& err);

extern const int kChannels[9];

}  // namespace hm
