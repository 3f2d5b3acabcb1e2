/*
 * p2p_mi355.h -- C ABI of libp2p_mi355.so: the MI355X (gfx950) drop-in for the Pix2Pose
 * inference hot path.  Plain pointers and sizes only; no torch / numpy / C++ types.
 *
 * Every entry point names the reference interface (kirumang/Pix2Pose, file:line) it
 * replaces.  All functions return P2P_OK (0) or a negative p2p_status; they never throw or
 * abort.  p2p_last_error() gives a human-readable reason for the calling thread.
 *
 * Threading: a p2p_ctx owns one HIP stream and its workspaces and is NOT thread-safe;
 * different contexts (one per GPU / rank) are independent.
 */
#ifndef P2P_MI355_H
#define P2P_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2P_ABI_VERSION 9

/* The library is built with -fvisibility=hidden: the entry points declared here (P2P_API) are its ONLY dynamic symbols
 * (tests/test_host_cpu.py holds `nm -D` to exactly this list). */
#if defined(__GNUC__)
#define P2P_API __attribute__((visibility("default")))
#else
#define P2P_API
#endif

typedef enum {
    P2P_OK = 0,
    P2P_ERR_INVALID_ARG = -1,   /* null pointer, bad size, unknown enum */
    P2P_ERR_HIP = -2,           /* a HIP runtime call failed (no GPU, OOM, launch failure) */
    P2P_ERR_WEIGHTS = -3,       /* missing / mis-sized weight tensor */
    P2P_ERR_CAPACITY = -4,      /* request exceeds a capacity fixed at ctx creation */
    P2P_ERR_RANGE = -5          /* a P2P_PREC_F16X3 generator pass stored an activation beyond the split-f16 operand range (see p2p_precision) */
} p2p_status;

/* reference: recognition.py:21-26 selects the generator graph from the `backbone` string */
typedef enum { P2P_BACKBONE_PAPER = 0, P2P_BACKBONE_RESNET50 = 1 } p2p_backbone;

/* where a caller buffer lives */
typedef enum { P2P_MEM_HOST = 0, P2P_MEM_DEVICE = 1 } p2p_mem;

typedef struct p2p_ctx p2p_ctx;     /* per-GPU pipeline context: stream + workspaces      */
typedef struct p2p_model p2p_model; /* one object's generator network, weights in HBM     */

/* One named weight tensor in Keras-native layout (see pix2pose_amd/weights.py):
 * Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin), Dense (in,out), BN vectors. */
typedef struct {
    const char* name;
    const float* data; /* host pointer */
    int64_t numel;
} p2p_tensor;

P2P_API int p2p_abi_version(void);
/* Binding self-checks: sizeof() of the public structs as this build of the library sees them
 * (which: 0 p2p_tensor, 1 p2p_image, 2 p2p_object, 3 p2p_detection, 4 p2p_pose, 5 p2p_est_pose_opts,
 * 6 p2p_kernel_stats; -1 otherwise), and the hash of the sources the library was built from
 * (pix2pose_amd/build.py) -- a foreign-language binding compares both with its own declarations / tree
 * before the first call, so that a stale .so is an error and not a silent struct mismatch. */
P2P_API int p2p_abi_sizeof(int which);
P2P_API const char* p2p_build_id(void);
/* Test hook, no GPU needed: the one-sided Gaussian weights (centre first) the anti-aliased resize uses for crop side
 * `side` against the 128-px network resolution; w must hold 256 doubles.  Returns the radius (0: no filtering), -1 on a bad
 * side.  tests/ hold it against scipy.ndimage's own kernel bit for bit. */
P2P_API int p2p_aa_weights(int side, double* w);
/* Test hook (GPU): the back-resizes of recognition.py:134,144,146 on caller-supplied 128 x 128 maps through the code the pipeline runs --
 * prob [128*128] (cval 1), pred [128*128*3] (img_pred, cval 0.5), non_gray [128*128] (0 / 1, cval 0), host float32 -- at out_h x out_w;
 * generation = p2p_est_pose_opts.resize_anti_aliasing.  Outputs (host): q [out_h*out_w*3] = (resize(pred) * 255) truncated to uint8,
 * below [out_h*out_w] = resize(prob) < th_inlier, ng_out = resize(non_gray) > 0.9.  tests/test_external_vectors.py holds it against the
 * vectors of the real scikit-image 0.18.3. */
P2P_API int p2p_debug_back_resize(p2p_ctx* ctx, const float* prob, const float* pred, const float* non_gray, int out_h, int out_w,
                                  double th_inlier, int generation, unsigned char* q, unsigned char* below, unsigned char* ng_out);
P2P_API const char* p2p_last_error(void);
P2P_API int p2p_device_count(int* count);

/* Create a context on `device`.  `max_batch` = largest number of 128x128 network inputs
 * processed per pass (activation workspace is sized for it; larger requests are chunked).
 * The est_pose workspaces (correspondences, canvases, landing buffers) are grow-only and sized by the batches
 * that arrive: nothing about frame or crop sizes is fixed here. */
P2P_API int p2p_ctx_create(int device, int max_batch, p2p_ctx** out);
P2P_API void p2p_ctx_destroy(p2p_ctx* ctx);
P2P_API int p2p_ctx_synchronize(p2p_ctx* ctx);
/* HIP stream handle (hipStream_t) the context launches on, for callers that time kernels
 * with HIP events or chain their own work. */
P2P_API void* p2p_ctx_stream(p2p_ctx* ctx);

/* Replaces `ae.aemodel_unet_*(p=1.0)` + `generator_train.load_weights(weight_fn)`
 * (reference recognition.py:21-26; graphs ae_model.py:70-150,175-240): uploads the
 * tensors, folds BatchNorm into per-channel scale/shift and re-packs kernels for the
 * MFMA implicit-GEMM kernels.  All tensors of pix2pose_amd.weights.tensor_specs(backbone)
 * must be present. */
P2P_API int p2p_model_create(p2p_ctx* ctx, const p2p_tensor* tensors, int n_tensors, int backbone,
                     p2p_model** out);

/* Arithmetic of the generator's dense contractions (the reference computes them in fp32 through
 * TensorFlow).  P2P_PREC_F32: fp32 matrix instructions, bitwise an fmaf chain.  P2P_PREC_F16X3: fp32
 * emulated on the f16 matrix pipe -- every operand is split into two f16 halves (22 significant
 * bits), three MFMAs per product block, fp32 accumulation; ~2.7x faster on the large layers, output
 * differs from the fp32 mode by ~1e-6 and is measured as close to a double-accumulating reference as
 * the fp32 mode is (max 2.5e-5 vs 3.3e-5 on the tanh outputs).  Weights are pre-scaled per output channel.
 * OPERAND RANGE of the split: |activation| < 65504 (f16 max).  Batch-normalised layers sit orders of magnitude below; the two
 * linear Dense layers (ae_model.py:199-200) have no BatchNorm behind them.  The range is GUARDED: every layer epilogue tracks the
 * largest magnitude it stores and raises a device-side flag beyond 6e4 -- a NaN counts as beyond (the running maximum and the ReLU of the
 * epilogues propagate NaN); the caller-supplied input x of p2p_predict / p2p_forward_async is NOT checked (|x| < 65504 is the caller's to
 * keep: the pipeline's own inputs are (u8 - 128) / 128) -- the flag travels with the results and
 *   p2p_predict / p2p_est_pose_batch / p2p_est_pose_collect return P2P_ERR_RANGE (outputs / poses of that call are not to be used);
 *   p2p_forward_async cannot report -- ask p2p_ctx_range_event() after synchronising.
 * P2P_PREC_AUTO: split-f16 with a strict-fp32 twin of the same weights kept beside it (+ the model's size in HBM).  On a range event the
 * object switches to the twin for good: p2p_predict and p2p_est_pose_batch repeat the work in fp32 themselves and return P2P_OK;
 * p2p_est_pose_collect returns P2P_ERR_RANGE once and the re-submitted batch runs in fp32.  The flag is per batch, not per object: in a
 * mixed batch every P2P_PREC_AUTO object switches, objects without a twin keep their arithmetic (if one of THOSE overflowed, the repeated
 * batch reports P2P_ERR_RANGE again).  p2p_model_precision() tells which
 * arithmetic an object currently uses.  p2p_model_create uses P2P_PREC_DEFAULT. */
typedef enum { P2P_PREC_F32 = 0, P2P_PREC_F16X3 = 1, P2P_PREC_AUTO = 2 } p2p_precision;
#define P2P_PREC_DEFAULT P2P_PREC_F16X3
P2P_API int p2p_model_create_ex(p2p_ctx* ctx, const p2p_tensor* tensors, int n_tensors, int backbone,
                        int precision, p2p_model** out);
P2P_API void p2p_model_destroy(p2p_model* model);
/* P2P_PREC_F32 or P2P_PREC_F16X3: the arithmetic the object's next pass will use (a P2P_PREC_AUTO object reports F16X3 until a
 * range event switched it to its fp32 twin); negative on a null handle. */
P2P_API int p2p_model_precision(const p2p_model* model);
/* Operand-range events of direct forward calls (p2p_forward_async) since the last query: synchronises the context stream, stores the
 * largest offending magnitude in *max_abs (0 = none) and clears the flag.  No reference counterpart (TensorFlow computes in fp32). */
P2P_API int p2p_ctx_range_event(p2p_ctx* ctx, float* max_abs);

/* Form of the 5x5 layers of the generator in P2P_PREC_F16X3 passes of this context: the stride-1 decoder convolutions deconv1 / deconv2 /
 * deconv3 (reference ae_model.py:207-211,217-220,227-230; 67 % of the generator's multiplications), the transposed convolutions up1 / up2 /
 * up3 (ae_model.py:201-204,212-215,222-225) and the stride-2 convolution conv4 (ae_model.py:190-195).  The Winograd forms along the row
 * axis -- F(4,5) for the stride-1 layers (2.5x fewer matrix-core products), F(4,3) on the sub-pixel phases / parity planes of the stride-2
 * layers (1.67x fewer) -- form different products than the direct convolutions, so the forms do not give the same bits: all are held to the
 * same bar against the oracle (network output within 1e-4; measured 4e-5 with every layer in Winograd form, 2.6e-5 direct;
 * tests/test_wino_gpu.py), 6e-5 apart at most.
 *   P2P_WINOGRAD_AUTO (default)  the fastest form at every pass size: F(4,5) for the stride-1 layers at every size, F(4,3) for up2 / up3 from 5
 *                                and for conv4 / up1 from 9 inputs; launches that would occupy a fraction of the chip split K over more
 *                                workgroups and add the partial sums in a separate step, in a fixed order (a pass over a few inputs: conv4
 *                                on the direct kernel, the stride-1 layers over ranges of channel slices, deconv1 up to 64 inputs; conv4 in
 *                                F(4,3) form over its four parity planes while under one workgroup per CU).  A sample's bits then depend on
 *                                the SIZE of the pass it travels in (never on its content, its order or the run)
 *   P2P_WINOGRAD_OFF             direct forms always   } either way a sample's output bits do not depend on the pass it travels in
 *   P2P_WINOGRAD_ALWAYS          Winograd forms always }  (conv4 split over its parity planes at every size)
 * Strict-fp32 objects (P2P_PREC_F32, the twin of P2P_PREC_AUTO) always use the direct forms.  No reference counterpart. */
typedef enum { P2P_WINOGRAD_OFF = 0, P2P_WINOGRAD_AUTO = 1, P2P_WINOGRAD_ALWAYS = 2 } p2p_winograd_mode;
P2P_API int p2p_ctx_set_winograd(p2p_ctx* ctx, int mode);

/* Replaces `self.generator_train.predict(x)` (reference recognition.py:84,129):
 * x [n,128,128,3] float32 NHWC -> xyz [n,128,128,3] (tanh) and prob [n,128,128,1] (sigmoid).
 * `mem` says whether x/xyz/prob are host or device pointers.  Blocking. */
P2P_API int p2p_predict(p2p_ctx* ctx, const p2p_model* model, const float* x, int n, float* xyz,
                float* prob, int mem);

/* Same forward pass, device buffers only, asynchronous on the context stream, output
 * interleaved [n,128,128,4] = (x,y,z,prob).  Used by the pipeline and by bench.py. */
P2P_API int p2p_forward_async(p2p_ctx* ctx, const p2p_model* model, const float* x_dev, int n,
                      float* xyzp_dev);


/* ------------------------------------------------------------------------------------------
 * Pose estimation: replaces `pix2pose.est_pose(rgb, bbox)` (reference recognition.py:70-193,
 * crop geometry get_boxes :28-69, pnp_ransac :195-224) for a whole batch of detections.
 * ---------------------------------------------------------------------------------------- */
#define P2P_MAX_OUTLIER_TH 8
/* Largest RANSAC iteration count (the reference hard-codes iterationsCount=100, recognition.py:217): the
 * hypothesis storage of the solver is sized for it; larger requests fail with P2P_ERR_INVALID_ARG. */
#define P2P_MAX_RANSAC_ITERATIONS 128

typedef enum { P2P_IMG_U8 = 0, P2P_IMG_F32 = 1 } p2p_img_dtype;

/* One H x W x 3 interleaved RGB frame (`rgb` of est_pose; uint8, or float32 as the ICP script
 * passes, reference tools/5_evaluation_bop_icp3d.py:369-370). */
typedef struct {
    const void* data;
    int height, width;
    int dtype; /* p2p_img_dtype */
    int mem;   /* p2p_mem */
} p2p_image;

/* Per-object state of a reference `pix2pose` instance (recognition.py:10-26): the generator
 * network plus obj_param = [x_scale,y_scale,z_scale,x_ct,y_ct,z_ct] (tools/bop_io.py:33-42),
 * th_outlier (1..8 values, evaluated in order), th_inlier, box_size. */
typedef struct {
    const p2p_model* model;
    double obj_scale[3];
    double obj_ct[3];
    int n_outlier_th;
    double outlier_th[P2P_MAX_OUTLIER_TH];
    double inlier_th;
    double box_size;
} p2p_object;

/* One 2D detection: which frame, which object, bbox = [v_min,u_min,v_max,u_max] (ints, as the
 * detectors hand them over, tools/5_evaluation_bop_basic.py:289-304) and the camera matrix the
 * caller would have assigned to `.camK` before the call (:302). */
typedef struct {
    int image;
    int object;
    int bbox[4];
    double camK[9];
} p2p_detection;

/* est_pose status: the reference signals failure in-band with -1 sentinels (recognition.py:79,
 * 127,191); the shim maps any status != 0 back to those sentinels. */
typedef enum {
    P2P_POSE_RANGE = -2,           /* gathered batches only: a detection of a rank whose batch returned P2P_ERR_RANGE -- it exists, its pose is not to be used */
    P2P_POSE_ABSENT = -1,          /* padding record of a gathered batch (p2p_est_pose_collect_gathered): no detection here */
    P2P_POSE_OK = 0,
    P2P_POSE_CROP_TOO_SMALL = 1,   /* recognition.py:78-79  */
    P2P_POSE_NO_CANDIDATE = 2,     /* recognition.py:125-127 */
    P2P_POSE_PNP_FAILED = 3        /* recognition.py:189-191 */
} p2p_pose_status;

typedef struct {
    double R[9];          /* rot_pred, row-major                           (recognition.py:223) */
    double t[3];          /* tra_pred, mm                                                       */
    double frac_inlier;   /* max_inlier / n_init_mask                      (recognition.py:193) */
    int n_inliers;        /* len(inliers) of the selected candidate                             */
    int n_init_mask;      /* stage-1 non-gray pixel count                  (recognition.py:90)  */
    int status;           /* p2p_pose_status                                                    */
    int best_slot;        /* index into outlier_th of the selected candidate, -1 if none        */
    int bbox_t[4];        /* [v1,v2,u1,u2] as returned by the reference (box of the LAST candidate) */
    int n_candidates;     /* stage-2 candidates that were built                                 */
    int ransac_iters;     /* RANSAC iterations run for the selected candidate                   */
    int64_t mask_stats[3];/* score_type 2 (p2p_est_pose_opts.det_mask given; tools/5_evaluation_bop_basic.py:307-316): {intersection, union,
                           * valid_mask pixel count} of the detector mask with valid_mask_full -- the same numbers opts.mask_stats receives,
                           * carried in the record so that a GATHERED record (p2p_est_pose_collect_gathered) is all another rank needs to
                           * score the detection; zeros otherwise */
} p2p_pose;

/* Optional knobs; zero-initialise for the reference behaviour. */
typedef struct {
    /* PnP-RANSAC constants hard-coded at recognition.py:216-217 / OpenCV defaults. 0 => default */
    int ransac_iterations;      /* 100; at most P2P_MAX_RANSAC_ITERATIONS */
    double reprojection_error;  /* 5.0  */
    double confidence;          /* 0.99 */
    /* TEST / BENCH ONLY: replace the decoder outputs after each generator pass (device
     * pointers).  inject1: [n_det,128,128,4]; inject2: [n_det,n_outlier_th_max,128,128,4].
     * The generator still runs (and is timed); its output is then overwritten.  No trained
     * weights exist offline, so synthetic scenes drive the PnP stage this way. */
    const float* inject1;
    const float* inject2;
    int inject_slots;           /* second dimension of inject2 */
    /* optional per-detection outputs, host pointers (may be null):
     * valid_mask [n_det][mask_stride] bytes (H*W of the detection's frame used),
     * img_pred   [n_det][pred_stride] bytes (crop h*w*3 used) */
    unsigned char* valid_mask;
    int64_t mask_stride;
    unsigned char* img_pred;
    int64_t pred_stride;
    /* optional debug taps (host pointers, may be null): stage-1 inputs [n_det,128,128,3] and
     * stage-2 inputs [n_det,K,128,128,3], K = max n_outlier_th over the objects in the batch */
    float* dbg_x1;
    float* dbg_x2;
    int* dbg_boxes2;            /* [n_det][12] stage-2 get_boxes result */
    int* dbg_cand;              /* [n_det][K][6]: valid, n_non_gray, n_corr, n_inliers, ransac iters, best iter */
    /* the generator's raw answers (x, y, z, prob) of both stages: [n_det,128,128,4] and [n_det,K,128,128,4] (slot k of a detection = its
     * k-th outlier threshold; slots without a stage-2 input hold garbage).  The shim builds the reference's FAILURE returns from them: the
     * first tuple element of recognition.py:127 / :191 is the stage-1 / last candidate's clipped (decode + 1) / 2 preview. */
    float* dbg_y1;
    float* dbg_y2;
    /* score_type 2 support (reference tools/5_evaluation_bop_basic.py:307-316): per-detection
     * detector masks, host pointer [n_det][det_mask_stride] bytes (non-zero = object), same H x W
     * as the detection's frame; mask_stats (host, [n_det][3]) receives
     * {intersection, union, valid_mask pixel count} of the detector mask with valid_mask_full. */
    const unsigned char* det_mask;
    int64_t det_mask_stride;
    int64_t* mask_stats;
    /* skimage.transform.resize GENERATION switch.  The reference calls resize(order=1) six times per detection
     * (recognition.py:82,103,121,134,144,146) and does not pin scikit-image (requirements.txt does not list it):
     * Values outside 0 .. 2 are refused (P2P_ERR_INVALID_ARG).
     *   0 (default): scikit-image <= 0.14 -- plain bilinear warp, every image warped in double (the 0.14 _warp_fast takes doubles only).
     *      PINNED to the real library's float64 warp: no 0.14 wheel exists in the build image, but what 0.14 runs for order=1 -- _warp_fast in
     *      double, the reflect / constant border modes, clip -- is what the real 0.18.3 runs for resize(image.astype(float64),
     *      anti_aliasing=False); tests/golden/reference_est_pose_skimage014.json is the reference's est_pose with that on all six call
     *      sites (masks, uint8 images, boxes, poses identical).  Restated: that 0.14 converts every image to double before the warp.
     *   1: scikit-image 0.17 - 0.18, PINNED bit for bit to the real 0.18.3 (+ scipy 1.7.1) of the build image's /opt/conda/bin/python3.9
     *      (tests/golden/external_vectors.json: resize itself; tests/golden/reference_est_pose_skimage018.json: the reference's est_pose
     *      with the real library on all six call sites -- masks, uint8 images, boxes identical):
     *      - anti_aliasing is on by default for float images: every DOWN-scaling resize of one is preceded by
     *        scipy.ndimage.gaussian_filter(sigma = (in/out - 1)/2, truncate 4, border 'mirror' / 'constant'+cval, result kept in the
     *        array's dtype), restated in csrc/resize_aa.hip -- and off for BOOL images (the keep mask of recognition.py:103);
     *      - warp() keeps a FLOAT32 image float32: the prob map of :134 and img_pred of :144 are interpolated by
     *        _warp_fast[float32] (matrix, source coordinates and taps in float32, the blend as the 0.18.3 wheel compiles it), the
     *        result is compared with th_inlier in float32 (:203) and multiplied by 255 in float32 (:144) before the uint8 truncation.
     *      One thing of the real library is NOT reproduced because it is not reproducible: resize() fits its scale-and-shift matrix by SVD,
     *      and the fit's 1e-14 noise -- which depends on the BLAS kernels the machine selects -- decides `> 0.9` on mask pixels whose
     *      bilinear weight is exactly 0.9 (crop sides that are multiples of 10).  This library uses the exact map the fit approximates;
     *      the fixture records both and shows the same wheels disagreeing with themselves across OPENBLAS_CORETYPE settings.
     *   2: scikit-image 0.15 / 0.16 -- what the reference's OWN image resolves to (requirements.txt:7 imgaug==0.2.7 pulls scikit-image,
     *      Dockerfile:1,5 is python 3.5, which caps it at 0.15.x).  anti_aliasing defaults to True for EVERY image: resize() hands the image
     *      as passed to scipy.ndimage.gaussian_filter, so the float32 maps are filtered in float32 as in generation 1 -- and the BOOL keep mask
     *      of recognition.py:103 is filtered into a BOOL array (each axis pass ends in a cast to npy_bool: a pixel survives only where its
     *      weighted sum reaches 1.0; depending on how the weights' sum rounds for the crop side, part of the mask survives or none of it);
     *      warp() then converts every image to double (the 0.15 / 0.16 _warp_fast takes doubles only), so interpolation, the th_inlier
     *      comparison and the * 255 are in double as in generation 0.  PINNING: no 0.15 wheel exists in the build image; the fixture
     *      (tests/golden/reference_est_pose_skimage015.json) runs the reference's est_pose with resize COMPOSED of the real scipy 1.7.1 filter on
     *      the image as passed and the real scikit-image 0.18.3 float64 warp -- filter (bool path included) and warp pinned, their order and the
     *      double conversion restated from the 0.15 / 0.16 sources.  The Gaussian weights are built with libm's exp like numpy <= 1.18 (the
     *      reference's era); numpy >= 1.19 uses a SIMD exp 1 ulp apart on some arguments, which the bool filter can amplify into a whole mask.
     *      So against a caller's scipy this generation is bit-exact only for crop sides whose weights agree between the two exp
     *      implementations (the parity tests skip the others; p2p_aa_weights() returns the library's weights of a side for comparison).
     * (scikit-image >= 0.19 rejects the bool array of recognition.py:103, so the reference does not run there.)
     * clip=True of resize (output clamped to the input's range, cval preserved) is common to all versions and always on. */
    int resize_anti_aliasing;
    /* p2p_est_pose_submit only: let the NEXT submit run this batch's stage-2 generator pass merged with its own stage-1
     * pass (one pass over [x2(k) | x1(k+1)] when the newcomer is not larger; collect runs it alone otherwise).  Fewer, larger
     * passes -- pays for small batches; at 256 detections per batch the chip is full either way (measured equal), so the
     * default is off.  Results are identical. */
    int merge_stream_passes;
    /* The valid_mask buffers are zero-filled already (freshly calloc'ed pages, numpy.zeros): the library then writes only the rows of
     * each detection's crop instead of clearing H x W bytes per detection first (256 detections of 640 x 480 frames: 79 MB of page
     * touching per batch on the calling thread).  0 = the library clears them (valid_mask_full = zeros, recognition.py:175).
     * PRECONDITION, not checked: with 1 the buffer must be all-zero before EVERY call that names it -- the library writes the crop rows
     * of detections with status OK and nothing else, so a buffer reused across calls (the natural pattern with submit / collect) keeps
     * the previous batch's rows, and the rows of detections that fail this time.  Re-zero it, allocate a fresh one, or pass 0. */
    int mask_prezeroed;
} p2p_est_pose_opts;

/* Blocking.  poses[i] corresponds to dets[i]. */
P2P_API int p2p_est_pose_batch(p2p_ctx* ctx, const p2p_object* objects, int n_objects, const p2p_image* images,
                       int n_images, const p2p_detection* dets, int n_dets, p2p_pose* poses,
                       const p2p_est_pose_opts* opts);

/* Asynchronous form of p2p_est_pose_batch for detection streams: `submit` only enqueues the batch
 * (generator passes + glue on the context stream, PnP-RANSAC + selection + D2H on a second stream)
 * and returns a ticket; `collect` waits for that batch and fills poses[n_dets].  At most two
 * batches may be in flight, so the latency-bound PnP tail of batch i overlaps the generator
 * passes of batch i+1.  Device-resident frames and injected maps must stay alive until the
 * collect.  The optional outputs of the reference's return tuple (valid_mask, img_pred: recognition.py:189-193)
 * and the score_type-2 sums (det_mask -> mask_stats) are available here too: name the host buffers in the
 * options of `submit`, keep them alive, and `collect` fills them (rendered on the tail stream, landed in pinned
 * memory, copied out at collect time).  det_mask is read during `submit`.  Only the debug taps are blocking-only. */
P2P_API int p2p_est_pose_submit(p2p_ctx* ctx, const p2p_object* objects, int n_objects, const p2p_image* images,
                        int n_images, const p2p_detection* dets, int n_dets, const p2p_est_pose_opts* opts,
                        int* ticket);
P2P_API int p2p_est_pose_collect(p2p_ctx* ctx, int ticket, p2p_pose* poses);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (no reference call site -- tools/5_evaluation_bop_basic.py:289-304 walks the detections of an
 * image one by one on one GPU).  Detections are independent: one process per GPU, each with its own p2p_ctx, runs the whole
 * pipeline on its shard.  The only exchange is one RCCL all-gather (xGMI) of the final p2p_pose records per batch.
 *   p2p_comm_unique_id   rank 0 draws the 128-byte id; the host program hands it to the other ranks by whatever it has
 *                        (MPI, a file, torch.distributed, a socket).
 *   p2p_comm_create      every rank: joins the communicator on its context's device (ncclCommInitRank -- collective).
 *   p2p_est_pose_collect_gathered   p2p_est_pose_collect + the gather: once the batch's tail has finished, its records go DEVICE to DEVICE over
 *                        the communicator (they never visit the host first) and are copied out once.
 *                        `poses[n_dets]` = this rank's own results as from collect; `gathered[world][n_max]` = every rank's records in
 *                        its caller's detection order, padded with status = P2P_POSE_ABSENT.  n_max >= the largest batch of any rank
 *                        and equal on all ranks.  Collective: every rank calls it once per step, in the same order.
 *                        A rank WITHOUT work this step (an empty shard: images run out on different ranks, an object group smaller than
 *                        the world -- the ragged loop of tools/5_evaluation_bop_basic.py:289-304) calls it with ticket = P2P_TICKET_NONE
 *                        (poses may be NULL) and contributes n_max padding records; p2p_est_pose_submit itself takes n_dets >= 1.
 *                        Errors never strand the peers: once ctx / comm / gathered / n_max are sane the all-gather is entered on every
 *                        path -- an unknown ticket, a batch larger than n_max (its ticket stays in flight for a plain collect) or a failed
 *                        stage-2 flush contribute padding and are returned AFTER the collective.  A batch that left the split-f16 operand
 *                        range (P2P_ERR_RANGE on this rank, nothing handed over) travels as P2P_POSE_RANGE records.
 * RCCL is bound at run time (an RCCL already mapped into the process -- PyTorch's wheel carries one -- is reused, else librccl.so.1;
 * P2P_RCCL_LIB overrides): the library loads and runs single-GPU on machines without RCCL; these calls then fail with P2P_ERR_HIP.
 * ---------------------------------------------------------------------------------------- */
#define P2P_COMM_ID_BYTES 128
#define P2P_TICKET_NONE (-1)
typedef struct p2p_comm p2p_comm;
P2P_API int p2p_comm_unique_id(char* id /* [P2P_COMM_ID_BYTES] */);
P2P_API int p2p_comm_create(p2p_ctx* ctx, int rank, int world, const char* id /* [P2P_COMM_ID_BYTES] */, p2p_comm** out);
P2P_API void p2p_comm_destroy(p2p_comm* comm);
P2P_API const char* p2p_comm_library(void);   /* path of the RCCL the calls above bound ("" if none could be) */
P2P_API int p2p_est_pose_collect_gathered(p2p_ctx* ctx, p2p_comm* comm, int ticket, p2p_pose* poses, int n_max, p2p_pose* gathered);

/* Replaces `cv2.solvePnPRansac(obj, img, camK, None, flags=EPNP, reprojectionError, iterationsCount)`
 * + `cv2.Rodrigues` (reference recognition.py:216-223) for a batch of independent problems.
 * Problem p owns points [offsets[p], offsets[p+1]) of obj_pts [N,3] (mm, double) and img_pts
 * [N,2] (pixels, double); camK [n_problems][9].  Host pointers.  ok[p]=0 reproduces cv2 returning
 * inliers=None.  inlier_mask (may be null) is [N] bytes.  info[p] = {n_inliers, iterations, best_iter}. */
P2P_API int p2p_pnp_ransac_batch(p2p_ctx* ctx, const double* camK, const double* obj_pts, const double* img_pts,
                         const int* offsets, int n_problems, int iterations, double reprojection_error,
                         double confidence, double* R, double* t, int* info, int* ok,
                         unsigned char* inlier_mask);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): when enabled, every launch of a convolution kernel of the generator is
 * bracketed by HIP events on the stream it is launched on; stats are per kernel family:
 *   0  igemm_kernel 128x128 tiles     1  igemm_kernel 128x64 tiles     2  igemm_kernel 128x32 tiles
 *   3  igemm_halo_kernel<2,2> (halo-tiled stride-1 multi-tap layers, 128x128 tiles)     4  igemm_halo_kernel<4,2> (Cout = 64 layers, 256x64 tiles)
 *   5  heads_halo_kernel (merged output heads)     6  igemm_halo8_kernel (layers on the 8x8 grid: conv4, first transposed conv)
 *   7  igemm_halo_s2_kernel (5x5 stride-2 convolutions on larger grids: the paper encoder's conv2 / conv3)
 *   8  igemm_stream_kernel (small launches: one wave per 32x32 tile, same K order and bits as the batched kernels)
 *   9  resblock_kernel (a ResNet identity bottleneck block -- 1x1, 3x3, 1x1 + residual -- in one launch, intermediates in LDS)
 *   10 wino_gemm_kernel (the 5x5 stride-1 decoder layers in Winograd F(4,5) form along the row axis: 2.5x fewer MFMA products; algo_flops
 *      stays the DIRECT form's 2 x MACs of the layer)     11 wino_input_kernel (its input transform: x -> split-f16 V in HBM; algo_flops 0)
 *   12..19 the bandwidth- / latency-leaning kernels around the generator, time and launches only (their work depends on device-side counts:
 *      bench.py prices them on the batch's correspondence / pixel counts):  12 pnp_hypotheses_kernel   13 pnp_count_kernel   14 pnp_score_kernel
 *      15 pnp_fit_solve_kernel + pnp_fit_select_kernel   16 aa_filter_kernel<0>   17 aa_filter_kernel<1>
 *      18 cand_eval_kernel + cand_compact_kernel (or cand_corr_kernel)   19 stage2_input_kernel
 *   20 wino3_gemm_kernel (the transposed convolutions up2 / up3 in Winograd F(4,3) form along the row axis: 15 position-products per input
 *      pixel instead of 25; algo_flops = the DIRECT form's 2 x MACs)     21 wino3_input_kernel (its input transform; algo_flops 0)
 * algo_flops counts the layers' algorithmic FLOPs (2 x MACs of the reference layer, SURVEY.md
 * section 8a-L), not padded work and not the 3 MFMA products per MAC of the split-f16 arithmetic.
 * ---------------------------------------------------------------------------------------- */
#define P2P_PROFILE_SLOTS 22
typedef struct {
    int64_t launches;
    double total_ms;
    double algo_flops;
    double algo_bytes;   /* compulsory HBM bytes of the launches: the layer's input tensor(s) + residual + weights read once, its output
                          * written once, fp32 -- what an HBM roofline of the bandwidth-leaning kernels is priced on */
} p2p_kernel_stats;

P2P_API int p2p_profile_enable(p2p_ctx* ctx, int on);
/* Harvest finished events (synchronises the stream) and return the accumulated stats; reset
 * clears the accumulators afterwards.  stats must hold P2P_PROFILE_SLOTS entries. */
P2P_API int p2p_profile_read(p2p_ctx* ctx, p2p_kernel_stats* stats, int reset);

#ifdef __cplusplus
}
#endif
#endif /* P2P_MI355_H */
