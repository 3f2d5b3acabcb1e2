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

#define P2P_ABI_VERSION 1

typedef enum {
    P2P_OK = 0,
    P2P_ERR_INVALID_ARG = -1,   /* null pointer, bad size, unknown enum */
    P2P_ERR_HIP = -2,           /* a HIP runtime call failed (no GPU, OOM, launch failure) */
    P2P_ERR_WEIGHTS = -3,       /* missing / mis-sized weight tensor */
    P2P_ERR_CAPACITY = -4       /* request exceeds a capacity fixed at ctx creation */
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

int p2p_abi_version(void);
const char* p2p_last_error(void);
int p2p_device_count(int* count);

/* Create a context on `device`.  `max_batch` = largest number of 128x128 network inputs
 * processed per pass (activation workspace is sized for it; larger requests are chunked).
 * `max_image_side` bounds the crop side used to size the PnP correspondence workspace. */
int p2p_ctx_create(int device, int max_batch, p2p_ctx** out);
void p2p_ctx_destroy(p2p_ctx* ctx);
int p2p_ctx_synchronize(p2p_ctx* ctx);
/* HIP stream handle (hipStream_t) the context launches on, for callers that time kernels
 * with HIP events or chain their own work. */
void* p2p_ctx_stream(p2p_ctx* ctx);

/* Replaces `ae.aemodel_unet_*(p=1.0)` + `generator_train.load_weights(weight_fn)`
 * (reference recognition.py:21-26; graphs ae_model.py:70-150,175-240): uploads the
 * tensors, folds BatchNorm into per-channel scale/shift and re-packs kernels for the
 * MFMA implicit-GEMM kernels.  All tensors of pix2pose_amd.weights.tensor_specs(backbone)
 * must be present. */
int p2p_model_create(p2p_ctx* ctx, const p2p_tensor* tensors, int n_tensors, int backbone,
                     p2p_model** out);
void p2p_model_destroy(p2p_model* model);

/* Replaces `self.generator_train.predict(x)` (reference recognition.py:84,129):
 * x [n,128,128,3] float32 NHWC -> xyz [n,128,128,3] (tanh) and prob [n,128,128,1] (sigmoid).
 * `mem` says whether x/xyz/prob are host or device pointers.  Blocking. */
int p2p_predict(p2p_ctx* ctx, const p2p_model* model, const float* x, int n, float* xyz,
                float* prob, int mem);

/* Same forward pass, device buffers only, asynchronous on the context stream, output
 * interleaved [n,128,128,4] = (x,y,z,prob).  Used by the pipeline and by bench.py. */
int p2p_forward_async(p2p_ctx* ctx, const p2p_model* model, const float* x_dev, int n,
                      float* xyzp_dev);

#ifdef __cplusplus
}
#endif
#endif /* P2P_MI355_H */
