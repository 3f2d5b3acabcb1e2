// Device-side records of the est_pose pipeline (crop -> generator -> masks -> re-crop ->
// generator -> correspondences -> PnP-RANSAC -> selection).  Reference: recognition.py:28-224.
#pragma once
#include "model.h"

namespace p2p {

constexpr int MAX_TH = P2P_MAX_OUTLIER_TH;

// get_boxes() result (recognition.py:28-69)
struct Boxes {
    int v1_ori, v2_ori, u1_ori, u2_ori;   // unclipped square
    int v1, v2, u1, u2;                   // clipped to the frame
    int vv1, vv2, uu1, uu2;               // paste window inside the square canvas
};

// Per-detection constants, filled on the host.
struct DetInfo {
    const void* img;        // device pointer to the frame
    int H, W;
    int img_f32;            // 0: uint8, 1: float32
    int obj;
    int n_th;
    float th_o[MAX_TH];     // prob < th_o is evaluated in float32 (prob is a float32 array)
    double th_i;            // img_prob_ori < th_i: in float64 for scikit-image <= 0.14 (a float64 array), in float32 for 0.17 / 0.18 (the array stays float32)
    double box_size;
    double cx_o, cy_o;      // (bbox[3]+bbox[1])/2, (bbox[2]+bbox[0])/2   recognition.py:72-73
    double K[9];
    double obj_scale[3], obj_ct[3];
    Boxes b1;
    int ok1;                // 0 => recognition.py:78-79 early return
    long long corr_off;     // float offset of this detection's correspondence storage
    int corr_cap;           // points per candidate (stage-1 side squared)
    int aa;                 // scikit-image generation (p2p_est_pose_opts.resize_anti_aliasing): 1 = 0.17 / 0.18 -- anti-aliased resizes AND float32 images warped
                            // in float32; 2 = 0.15 / 0.16 -- anti-aliased resizes of EVERY image (the bool keep mask too), every image warped in double
    long long cv_off;       // double offset of this detection's (1 + K) canvases of corr_cap * 3 doubles each (aa, side > 128)
    int src_index;          // index of this detection in the caller's array (detections are processed sorted by object)
};

// One image of an anti-aliased resize (skimage: ndi.gaussian_filter before the warp): filtered in place.
struct AaItem {
    double* a;              // [H][W][C] interleaved
    double* tmp;            // scratch of the same size (result of the first axis pass)
    int H, W, C;
    int radius;             // 0 = nothing to do (item inactive, or sigma == 0)
    const double* w;        // one-sided weights w[0 .. radius], centre first
    int mode;               // 0 = 'mirror' (skimage mode 'reflect'), 1 = 'constant'
    int round32;            // the image is a float32 array in the reference: round to float32 after each axis pass
    double cval;
    int r0, r1, c0, c1;     // r1 > r0: region of interest -- rows [r0, r1) x columns [c0, c1) hold everything of the image that is not EXACTLY zero
                            // (stage-2 canvases: the object's silhouette is ~45 % of the crop); the image and its filter passes are neither written
                            // nor read outside the region (grown by the radius per pass), readers take +0.0 there.  r1 <= r0: the whole image
    double vmin, vmax;      // range of the filtered image (skimage clips the warp output to it)
    unsigned long long kmin, kmax;   // the same as order-preserving keys while the filter's workgroups are still reducing into them
};

// Ranges skimage's clip=True needs for the 'constant'-mode back-resizes of one candidate (recognition.py:134,144,146):
// [min, max] of the warp INPUT (the raw 128x128 map, or its anti-aliased version).
struct CandRange {
    double pmin, pmax;      // prob                       (cval 1)
    double qmin, qmax;      // img_pred, all 3 channels   (cval 0.5)
    double gmin, gmax;      // non_gray as float          (cval 0)
};

// Gaussian weights per crop side (resize_aa.hip), device pointers
struct AaTable {
    const double* w = nullptr;
    const int* off = nullptr;
    const int* rad = nullptr;
    int max_side = 0;
};
int aa_table_get(int device, AaTable* out);          // builds + uploads once per device
hipError_t launch_aa_filter(AaItem* items, int n_items, int max_elems, hipStream_t s);   // both axis passes + range


// Stage-1 reductions + stage-2 geometry, written by the device.
struct Stage1 {
    int n_init_mask;
    int bb[4];              // min v, min u, max v, max u of non_gray
    long long sum_v, sum_u;
    int keep_cnt[MAX_TH];
    int valid2[MAX_TH];     // candidate built for threshold slot k
    double kmin[MAX_TH], kmax[MAX_TH];   // range of the keep mask image fed to the warp (clip=True), per slot
    int keepf[MAX_TH];      // generation 2: the slot's keep mask was filtered (AaPtrs::keepf holds it)
    int n_cand;
    Boxes b2;               // all candidates of a detection share it (quirk)
};

// Per-candidate reductions written by the correspondence kernel.
struct CandStat {
    int n_non_gray;
    int n_corr;
    long long sum_v, sum_u; // of non_gray pixels, frame coordinates
};

struct PnpProblem {
    const float* pts;       // SoA: X[cap] Y[cap] Z[cap] U[cap] V[cap] (float32, as OpenCV stores them)
    int cap;
    int n;
    double K[9];
    unsigned char* mask;    // optional inlier mask [n]
};

struct PnpResult {
    double R[9];
    double t[3];
    int n_inliers;          // -1 on failure (recognition.py:215,219)
    int iters;
    int best_iter;
    int ok;
};

// workspace: pnp_workspace_bytes(n_problems) bytes of device memory (hypothesis models)
size_t pnp_workspace_bytes(int n_problems);
hipError_t launch_pnp_ransac(const PnpProblem* probs, PnpResult* results, int n_problems, int iterations,
                             double reproj_err, double confidence, int min_points, int max_points, double* workspace, hipStream_t s);
// max_points: upper bound of PnpProblem::n over the problems (shapes the counting launch; results do not depend on it)

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// growable pinned host buffer (D2H landing zone: the copy is truly asynchronous and the caller's pageable
// buffer is filled with one memcpy at collect time)
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// anti-aliased resizes (resize_aa.hip): descriptor arrays per image kind, all null when the option is off
struct AaPtrs {
    AaItem* k0;   // [n]        stage-1 canvases           (side > 128)
    AaItem* k1;   // [n*K]      stage-2 canvases           (side > 128)
    AaItem* k3;   // [n*K*5]    prob, pred r/g/b, non_gray (stage-2 side < 128)
    const unsigned char* keepf;   // generation 2 only: [n*K][128*128] keep masks after scipy's filter with bool output (stage-1 side < 128; Stage1::keepf says which)
};
struct AaBufs {
    double *cv, *cv_tmp;      // canvases: per detection (1 + K) x corr_cap x 3 doubles at DetInfo::cv_off
    double *bk, *bk_tmp;      // back-resize planes: [n*K][5][128*128]
};
struct BatchGroup { int obj, begin, end; };   // detections [begin, end) of the sorted batch belong to object `obj`

// Per-batch buffers: double buffered.  Asynchronous batches alternate between two generator lanes (stream +
// activation workspace, Ctx::lane[0/1]) so that the passes of batch i+1 fill the launch tails of batch i, and
// their PnP-RANSAC tails run on a third stream.
struct Slot {
    DevBuf det, s1, cand, probs, results, poses, corr, hyp;
    DevBuf x1, y1, x2, y2, images;      // network inputs / outputs of both stages, uploaded frames
    DevBuf crange;                      // CandRange per candidate
    DevBuf sacc;                        // few detections: global accumulators of the segmented stage-1 reductions (self-clearing)
    int sacc_n = 0;
    DevBuf crec, cseg;                  // few candidates: per-pixel records and per-segment counts of the two-launch correspondence build
    DevBuf aa_items, aa_cv, aa_cv_tmp, aa_bk, aa_bk_tmp;   // anti-aliased resizes: descriptors, canvases, 128x128 planes
    DevBuf keepf;                       // generation 2: filtered bool keep masks
    DevBuf mask, pred, dmask, mstat;    // optional outputs of the batch (valid_mask_full, img_pred_f, detector masks, IoU sums)
    p2p_pose* host_poses = nullptr;     // pinned
    size_t host_cap = 0;
    PinnedBuf h_mask, h_pred, h_stat;   // pinned landing buffers of the optional outputs (sorted order)
    PinnedBuf h_frames;                 // pinned staging of host frames (pageable caller memory -> here -> DMA)
    PinnedBuf h_range;                  // operand-range word of this batch's generator passes (kernels.h: RANGE_LIMIT), landed with the poses
    int range_word = 0;                 // index into Ctx::range_words (1 + slot index)
    unsigned* range_dev = nullptr;      // = Ctx::range_words + range_word
    // host-side description of the batch (kept from submit to collect: the stage-2 pass of an asynchronous batch is enqueued
    // by the NEXT submit, merged with that batch's stage-1 pass, or by collect)
    std::vector<int> perm;
    std::vector<DetInfo> hd;
    std::vector<BatchGroup> groups;
    std::vector<p2p_object> objs;
    int K = 0;
    bool identity = true, use_aa = false, same_backbone = true;
    bool stage2_pending = false;        // stage-2 inputs are built, the stage-2 generator pass and the tail are not enqueued yet
    int tail_cap = 0;                   // network inputs of a following batch that fit behind this batch's stage-2 inputs
    int max_side = 0;
    AaPtrs aa = {nullptr, nullptr, nullptr, nullptr};
    std::vector<int> img_hw, img_w;     // H*W and W of each detection's frame (sorted order)
    long long cmask_stride = 0, cpred_stride = 0;   // bytes per detection of the compact mask / image landing buffers
    p2p_est_pose_opts opt;              // the caller's output pointers (host), filled at collect time
    int n = 0;
    int ticket = -1;                    // in-flight async batch, -1 = free
    hipEvent_t done = nullptr;
};

struct Pipeline {
    static constexpr int N_SLOTS = 2;   // asynchronous batches in flight (each on its own generator lane)
    Slot slot[N_SLOTS];
    hipStream_t tail_stream = nullptr;  // PnP + selection + D2H of async batches
    hipStream_t copy_stream = nullptr;  // host frames -> HBM, under the previous batch's generator passes
    hipEvent_t corr_ready = nullptr;
    hipEvent_t frames_ready = nullptr;
    int next_ticket = 0;
    ~Pipeline();
};

// RCCL all-gather of a batch's pose records (comm.hip)
struct Comm;
int comm_world(const Comm& C);
int comm_gather(Ctx& X, Comm& C, Slot* s, bool range_event, hipStream_t ts, int n_max, p2p_pose* gathered);

}  // namespace p2p
