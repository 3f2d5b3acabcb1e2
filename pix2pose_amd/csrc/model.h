// Host-side objects behind the opaque handles of include/p2p_mi355.h.
#pragma once
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/p2p_mi355.h"
#include "kernels.h"

namespace p2p {

void set_error(const char* fmt, ...);
const char* get_error();

// One packed dense contraction (device pointers).
struct ConvLayer {
    float* w = nullptr;       // [Cout_pad][K]  (conv_first: [K][Cout])
    float* scale = nullptr;   // folded BatchNorm scale per cout
    float* shift = nullptr;   // bias / folded BatchNorm shift per cout
    int K = 0, Cout = 0, ntaps = 0;
    int prec = PREC_F32;      // arithmetic the panel `w` is packed for (Prec)
    std::string name;         // key in Model::L (grouped launches look the same layer up in every object)
    int8_t dy[IGEMM_MAX_TAPS + 3] = {0};
    int8_t dx[IGEMM_MAX_TAPS + 3] = {0};
    // 5x5 stride-1 layers of split-f16 models: the Winograd F(4,5) panel beside the direct one (wino.hip; kernels.h: WinoParams)
    float* wino_u = nullptr;
    float* wino_scale = nullptr;   // folded BatchNorm scale times the inverse of the Winograd panel's per-channel pre-scale
    size_t wino_bytes = 0;
    // 3x3 layers of the fused bottleneck blocks (split-f16 models): the panel `w` once more in MFMA fragment order (resblock.hip, phase B)
    float* w_frag = nullptr;
};

struct Model {
    int backbone = 0;
    int prec = PREC_F32;
    int device = 0;
    std::map<std::string, ConvLayer> L;
    std::map<std::string, float*> block_ss;   // identity bottleneck blocks (split-f16 models): folded BatchNorm of the block's three layers in one array (resblock.hip)
    // P2P_PREC_AUTO: a split-f16 model with a strict-fp32 twin of the same weights.  The generator runs split-f16 until an operand-range
    // event (kernels.h: RANGE_LIMIT) is seen on a pass of this object; from then on every pass of the object uses the twin.
    Model* twin = nullptr;
    mutable std::atomic<bool> use_twin{false};      // written through const handles when a range event is seen; a model may be shared between contexts / threads
    const Model* effective() const { return (use_twin && twin) ? twin : this; }
    ~Model();
};

struct Pipeline;   // est_pose workspaces (pipeline.hip)

// Objects sharing one generator pass (mixed batch, detections sorted by object): samples
// [start[g], start[g+1]) of the pass belong to models[g].
struct GroupCtx {
    std::vector<const Model*> models;
    std::vector<int> start;
};

struct Ctx {
    int device = 0;
    int max_batch = 0;
    hipStream_t stream = nullptr;         // == lane[0].stream
    // A lane = a HIP stream + its own activation workspace.  Lane 0 serves every single-object call;
    // the side lanes let the generator passes of DIFFERENT objects in a mixed batch run concurrently
    // (each is a small launch sequence that cannot fill 256 CUs alone; BASELINE.json configs[3]).
    static constexpr int N_LANES = 4;
    struct Lane {
        hipStream_t stream = nullptr;
        std::map<std::string, float*> act;    // activation workspace, sized for max_batch inputs
        hipEvent_t done = nullptr;
    };
    Lane lane[N_LANES];
    Lane* cur = &lane[0];                 // lane the next forward_chunk() runs on
    const GroupCtx* grp = nullptr;        // set while a grouped (multi-object) pass is being enqueued
    hipEvent_t fork = nullptr;
    int ensure_lane(int i);
    float* x_stage = nullptr;
    float* xyzp_stage = nullptr;
    float* xyz_stage = nullptr;
    float* prob_stage = nullptr;
    Pipeline* pipe = nullptr;
    int wino_mode = P2P_WINOGRAD_AUTO;    // p2p_ctx_set_winograd
    int dev_part = 0;                     // timing builds only (P2P_TIMING_SWITCHES): run a part of the generator pass
    // operand-range guard (kernels.h): device words raised by the epilogues of split-f16 passes.  Word 0: direct forward calls
    // (p2p_predict / p2p_forward_async), words 1.. : one per est_pose batch slot.  range_cur = where the passes being enqueued report.
    unsigned* range_words = nullptr;
    unsigned* range_cur = nullptr;
    int range_read(int word, float* out);      // synchronous read-and-clear of a word (after the stream that wrote it is idle)
    // measurement hooks (p2p_profile_*)
    bool profiling = false;
    struct ProfEvent { hipEvent_t a, b; int cfg; double flops, bytes; };
    std::vector<ProfEvent> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    p2p_kernel_stats prof_stats[P2P_PROFILE_SLOTS] = {};   // per kernel family (p2p_mi355.h)
    hipEvent_t prof_get_event();
    int prof_harvest();
    int ensure_workspace();
    void free_pipeline();
    ~Ctx();
};

// Sets the glue / PnP measurement hook (kernels.h: ProfHook) for the calling thread while the context is profiling.
struct ProfHookGuard {
    ProfHook prev;
    explicit ProfHookGuard(Ctx& X);
    ~ProfHookGuard();
};

int forward_chunk(Ctx& X, const Model& M, const float* x_dev, int n, float* xyzp_dev);
int forward_async(Ctx& X, const Model& M, const float* x_dev, int n, float* xyzp_dev);
// one generator pass over the detections of several objects (same backbone), sorted by object:
// every layer is ONE grouped launch in which each M-tile uses its object's weight panel
int forward_grouped(Ctx& X, const std::vector<const Model*>& models, const std::vector<int>& counts, const float* x_dev,
                    float* xyzp_dev);

}  // namespace p2p
