/*
 * conv3p.h -- C ABI of the MI355X-native conv3p operator pair (libconv3p_hip.so).
 *
 * This is the drop-in boundary for the hot path of hkust-vgd/pointwise: the two
 * TensorFlow custom ops `Conv3p` and `Conv3pGrad` of tf_conv3p.so
 *   schema      /root/reference/tf_ops/conv3p/register_op.cpp:44-75   (atrous variant)
 *   CPU kernels /root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp:401-509, :526-720
 *   GPU kernels /root/reference/tf_ops/conv3p/tf_conv3p_atrous.cu:541-642, :659-775  (replaced)
 *   callers     /root/reference/pointcnn2_acsd.py:10-31,
 *               /root/reference/scene_seg/pointcnn_scene_seg_acsd.py:9-30
 * A TensorFlow OpKernel shim (integration/tf_conv3p_shim.cc) or
 * the Python host mirror (pointwise_amd/conv3p_op.py) binds exactly these symbols.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     stated otherwise; all tensors dense row-major, dtype float (f32) or double (f64),
 *     the two dtypes the reference registers (register_op.cpp:45).
 *       points  (B, N, 3)            input  (B, N, Cin)
 *       filter  (fz, fy, fx, Cin, Cout)   weight index (f*Cin + k)*Cout + c,
 *                                         tap f = (tz*fy + ty)*fx + tx   (.cpp:290, :490)
 *       output / grad_out (B, N, Cout)    grad_input (B, N, Cin)   grad_filter like filter
 *   - stride_xyz is a HOST pointer to {sx, sy, sz} (the reference's int32[3] `stride`
 *     input, read on the host at .cpp:438-440); voxel_size is passed by value (the
 *     reference's T[1] `voxel_size` input, .cpp:444).  The non-atrous schema
 *     (register_op.cpp:9-38) is the special case stride = {1,1,1}.
 *   - outputs are fully overwritten: the library zero-initialises them itself
 *     (reference: memset at .cpp:451, :580, :590).
 *   - no allocation, no host synchronisation, no stdout in the hot calls: all scratch
 *     comes from the caller's `workspace` (size from conv3p_workspace_bytes), and every
 *     call is ordered on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - every function returns a status code (0 = OK); nothing throws or exits.
 *     CONV3P_ERR_INVALID_ARGUMENT corresponds to the reference's
 *     errors::InvalidArgument / OP_REQUIRES failures (.cpp:410-443, :549-585).
 *   - re-entrant and thread-safe for distinct workspaces.
 *   - 64-bit sizes internally: every element offset and byte count is size_t.
 */
#ifndef CONV3P_H
#define CONV3P_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONV3P_ABI_VERSION 5

/* status codes */
#define CONV3P_OK 0
#define CONV3P_ERR_INVALID_ARGUMENT 1 /* shape / stride / voxel validation failed            */
#define CONV3P_ERR_WORKSPACE 2        /* workspace NULL, misaligned (256 B) or too small     */
#define CONV3P_ERR_UNSUPPORTED 3      /* configuration outside what the kernels handle       */
#define CONV3P_ERR_LAUNCH 4           /* HIP reported an error when launching                */
#define CONV3P_ERR_NO_DEVICE 5        /* no gfx950-class HIP device visible                  */

/* which op the workspace is for */
#define CONV3P_PASS_FORWARD 0
#define CONV3P_PASS_BACKWARD 1
#define CONV3P_PASS_NEIGHBOR_COUNT 2

/* Bytes of scratch the given op needs for these shapes (elem_bytes = 4 or 8).
 * Returns 0 for invalid shapes.  Replaces the reference's per-call
 * OpKernelContext::allocate_temp (tf_conv3p_atrous.cu:97-106, :598-606). */
size_t conv3p_workspace_bytes(int pass, int elem_bytes, int B, int N, int Cin, int Cout, int fz,
                              int fy, int fx);

/* Conv3p forward.  Replaces Conv3pOp<Device,T>::Compute
 * (tf_conv3p_atrous.cpp:401-509 / tf_conv3p_atrous.cu:541-642). */
int conv3p_forward_f32(const float *points, const float *input, const float *filter,
                       const int32_t *stride_xyz, float voxel_size, int B, int N, int Cin,
                       int Cout, int fz, int fy, int fx, float *output, void *workspace,
                       size_t workspace_bytes, void *stream);
int conv3p_forward_f64(const double *points, const double *input, const double *filter,
                       const int32_t *stride_xyz, double voxel_size, int B, int N, int Cin,
                       int Cout, int fz, int fy, int fx, double *output, void *workspace,
                       size_t workspace_bytes, void *stream);

/* Conv3pGrad.  Replaces Conv3pGradOp<Device,T>::Compute
 * (tf_conv3p_atrous.cpp:526-720 / tf_conv3p_atrous.cu:659-775).
 * Input order follows the op schema (register_op.cpp:63-72): grad_from_next first. */
int conv3p_backward_f32(const float *grad_out, const float *points, const float *input,
                        const float *filter, const int32_t *stride_xyz, float voxel_size, int B,
                        int N, int Cin, int Cout, int fz, int fy, int fx, float *grad_input,
                        float *grad_filter, void *workspace, size_t workspace_bytes, void *stream);
int conv3p_backward_f64(const double *grad_out, const double *points, const double *input,
                        const double *filter, const int32_t *stride_xyz, double voxel_size, int B,
                        int N, int Cin, int Cout, int fz, int fy, int fx, double *grad_input,
                        double *grad_filter, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Neighbour cache (optional; not in the reference).
 *
 * Everything geometric the ops compute -- curve-sorted point records, per-tap populations and
 * the per-centre neighbour lists with their taps and normalisers -- depends only on `points` and
 * on the stencil (filter extents, stride, voxel_size), not on features, weights or gradients.
 * In the reference's models every layer of a step sees the same `points`, and Conv3pGrad repeats
 * Conv3p's search (pointcnn2_acsd.py:48-66; tf_conv3p_atrous.cu recomputes neighbor_count in both
 * ops).  The *_cached entry points keep that state in a caller-owned, persistent device buffer:
 *
 *   - `cache` must be zero-filled once before its first use and must not be written by the
 *     caller afterwards; one cache serves one (B, N, dtype) at a time (a change of shape simply
 *     invalidates it).  It also holds the per-call scratch, so no separate workspace is needed.
 *   - Validity is decided ON THE DEVICE, per cloud: a 64-bit content hash of the raw coordinates
 *     is recomputed every call (one light kernel) and compared with the hash the lists were built
 *     from; the search / finalise kernels are launched every call and return immediately for
 *     clouds whose lists are current.  There is no host synchronisation and no pointer-identity
 *     assumption: a recycled or overwritten buffer can only cost a rebuild.
 *   - `slots` stencils are kept at once (LRU): 4 for the classification stack, 5 for segmentation.
 *   - Results are bitwise identical to the stateless entry points.
 * The stateless conv3p_forward/backward_* run the very same kernels on `workspace` with the cache
 * logic forced to "rebuild".
 * ------------------------------------------------------------------------------------------- */
typedef struct conv3p_cache_config {
    int slots;           /* stencils cached simultaneously (1..64)                                   */
    int max_taps;        /* largest fz*fy*fx that will be used with this cache (27 for 3x3x3)        */
    int pairs_per_point; /* pair-list capacity per point, averaged over a cloud; 0 = default (256).
                            A cloud that needs more is still handled correctly (slow path).        */
    int max_Cin;         /* largest channel counts of a backward call (sizes the scratch for the   */
    int max_Cout;        /*   per-workgroup grad_filter partials).  Forward calls never need them:  */
                         /*   a forward whose faster variant wants scratch the cache lacks (the     */
                         /*   transform + gather forward of 36 -> 13) runs the plain kernel instead, */
                         /*   same results.  Wide (matrix-core) layers need a cache sized for them. */
    int flags;           /* CONV3P_CACHE_* bits below                                                */
} conv3p_cache_config;

/* Caller's promise for THIS call: `points` holds exactly the bytes it held at the previous *_cached_* call
 * on this cache (e.g. the later layers of a model step, all fed by one points tensor).  The library then
 * skips the content hash and, for a stencil it has already built since the last un-hinted call, the search
 * launches as well.  Without the flag every call re-validates on the device (always safe).  A wrong promise
 * gives results for the previous clouds. */
#define CONV3P_CACHE_POINTS_UNCHANGED 1
/* Which backward kernel serves the dilated narrow layers (9 -> 9 at strides 2-4) depends on the DATA: with short pair
 * lists (surface-sampled objects at N <= 2048: 7-27 neighbours per point) the one that keeps its G matrix for the
 * populated (centre, tap) rows only (37 KiB of LDS instead of 79: four workgroups per CU) is faster (cfg2: -7 % per
 * step); with long ones (S3DIS-like rooms, ~50 neighbours) it is 30 % slower than the dense-G kernel; measured
 * crossover between 27 and 41.  By DEFAULT the library decides on the device: the search leaves, per stencil, a word
 * saying whether the lists it just built are short on average (<= 35 pre-filter hits per point over the batch), the
 * backward launches BOTH kernels and that word lets exactly one of them run -- no host synchronisation, nothing for
 * the caller to know, one empty launch (~5 us) per such backward call.  A caller who knows its data can save that
 * launch with one of the two hints below, valid for the lifetime of the cache (Conv3pStack.tune() measures a sample
 * batch once at set-up and sets one).  Results are the reference's either way (same decisions, tolerance of the op);
 * the two kernels do not give bitwise the same sums, so runs that must reproduce each other bit for bit should fix the
 * choice with a hint or keep the data regime well away from the threshold. */
#define CONV3P_CACHE_SPARSE_NEIGHBOURHOODS 2   /* short lists: the populated-rows kernel alone */
#define CONV3P_CACHE_DENSE_NEIGHBOURHOODS 8    /* long lists: the dense-G kernel alone */
/* conv3p_cache_prepare_f32 only: besides the geometry, also build the two record orders (by forward tap, by backward
 * tap) that the matrix-core path of the wide layers (more than 16 channels on either side, fp32) derives from it -- about
 * 7 % of such a layer's forward+backward.  The next forward / backward on the same points in this cache (called with
 * CONV3P_CACHE_POINTS_UNCHANGED) then finds them in the cache's scratch region; any other use of the cache in between
 * (another stencil's wide layer, a narrow layer) simply rebuilds them.  Needs a cache sized for a wide layer
 * (max_Cin / max_Cout of conv3p_cache_config); CONV3P_ERR_WORKSPACE otherwise.  A performance hint only. */
#define CONV3P_CACHE_PREPARE_DEEP_ORDERS 4
/* conv3p_stack_forward_* / conv3p_stack_backward_* only, OPT-IN: run the hidden layers of a pass as ONE launch with
 * per-cloud barriers between the layers (pointwise_amd/csrc/conv3p_stack_fused.hpp) where the stack, the cache and the
 * device allow it (fp32, in_channels 3 or 9, hidden 9, the whole grid resident at once; the backward also needs
 * CONV3P_CACHE_SPARSE_NEIGHBOURHOODS).  One bit per pass.  Same bits for the activations and grad_input as the per-layer
 * launches; grad_filter sums its partials in another order (tolerance of the op).  Measured (profiles/r06_ab_fused.txt,
 * r06_cfg4_ab.txt): the fused FORWARD is 68 against 86 us on cfg2 and nothing runs beside the forward
 * in a pipelined step, so it is a gain there; the fused BACKWARD holds every slot of the chip while its tiles wait for the
 * slowest tile of their cloud, which keeps the next batch's search out (cfg2: no gain); on the rooms of cfg4, whose tiles
 * differ far more, both lose (1.34 against 1.26 ms).  Hence opt-in; Conv3pStack.tune() sets the forward bit for clouds
 * with short pair lists.  ONE fused launch at a time per device: a waiting tile needs the rest of its cloud resident or
 * dispatchable, and two fused launches issued concurrently (two streams, two processes on one GPU) can hold the slots each
 * other's tiles wait for; every wait is bounded (half a second), a wait that gives up is reported through
 * conv3p_cache_fused_status AND fails the next stack call on that cache with CONV3P_ERR_LAUNCH. */
#define CONV3P_CACHE_FUSED_FORWARD 16
#define CONV3P_CACHE_FUSED_BACKWARD 32
#define CONV3P_CACHE_FUSED_STACK (CONV3P_CACHE_FUSED_FORWARD | CONV3P_CACHE_FUSED_BACKWARD)

size_t conv3p_cache_bytes(int elem_bytes, int B, int N, const conv3p_cache_config *cfg);
/* Drop the host-side bookkeeping of a cache buffer (call before freeing it). */
int conv3p_cache_forget(void *cache);
/* Make `cache` (cache_bytes of freshly allocated, possibly RECYCLED device memory) a valid empty cache: zero-fills it on
 * `stream` and drops any host bookkeeping left for that address.  The "zero-filled once" requirement above, as a call:
 * a framework allocator (TensorFlow's BFC) can hand back a region that overlaps a cache freed earlier in the step, whose
 * hashes and slot marks may have survived while its lists were overwritten by temporaries -- content hashes cannot
 * notice that, zero-filling can (integration/tf_conv3p_shim.cc calls this after every allocate_persistent). */
int conv3p_cache_init(void *cache, size_t cache_bytes, void *stream);

int conv3p_forward_cached_f32(const float *points, const float *input, const float *filter,
                              const int32_t *stride_xyz, float voxel_size, int B, int N, int Cin,
                              int Cout, int fz, int fy, int fx, float *output, void *cache,
                              size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_forward_cached_f64(const double *points, const double *input, const double *filter,
                              const int32_t *stride_xyz, double voxel_size, int B, int N, int Cin,
                              int Cout, int fz, int fy, int fx, double *output, void *cache,
                              size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_backward_cached_f32(const float *grad_out, const float *points, const float *input,
                               const float *filter, const int32_t *stride_xyz, float voxel_size,
                               int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                               float *grad_input, float *grad_filter, void *cache,
                               size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_backward_cached_f64(const double *grad_out, const double *points, const double *input,
                               const double *filter, const int32_t *stride_xyz, double voxel_size,
                               int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                               double *grad_input, double *grad_filter, void *cache,
                               size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);

/* Build (or re-validate) the geometry of one stencil without running an op: a later forward / backward
 * with the same points and stencil finds its lists ready.  Lets a caller enqueue the searches of a model's
 * later layers on a second stream while the first layers' accumulation kernels run (the search is
 * VALU-bound, the accumulation gather-latency-bound; they overlap well). */
int conv3p_cache_prepare_f32(const float *points, const int32_t *stride_xyz, float voxel_size, int B,
                             int N, int fz, int fy, int fx, void *cache, size_t cache_bytes,
                             const conv3p_cache_config *cfg, void *stream);
int conv3p_cache_prepare_f64(const double *points, const int32_t *stride_xyz, double voxel_size, int B,
                             int N, int fz, int fy, int fx, void *cache, size_t cache_bytes,
                             const conv3p_cache_config *cfg, void *stream);

/* The same for n_strides stencils that share filter extents, voxel size and points (the layers of the
 * reference's models: strides 1..4, pointcnn2_acsd.py:47-65) with one sort, ONE search launch and ONE
 * normaliser launch: stride_xyz is int32[n_strides][3] on the host, n_strides <= 8 and <= cfg->slots.  A single
 * search launch is one round of workgroups whose duration is set by its slowest tile; batched, the stencils
 * fill each other's tails. */
int conv3p_cache_prepare_multi_f32(const float *points, const int32_t *strides_xyz, int n_strides,
                                   float voxel_size, int B, int N, int fz, int fy, int fx, void *cache,
                                   size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_cache_prepare_multi_f64(const double *points, const int32_t *strides_xyz, int n_strides,
                                   double voxel_size, int B, int N, int fz, int fy, int fx, void *cache,
                                   size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);

/* Per-point, per-tap neighbour populations, int32 (B, N, fz*fy*fx) on the device.
 * Restates Grid::neighbor_count / kernelBuildNeighborCount
 * (tf_conv3p_atrous.cpp:306-379 / tf_conv3p_atrous.cu:288-343): the intermediate both
 * ops normalise by.  Exported so that neighbour / tap decisions can be checked for
 * exact integer equality against the CPU reference. */
int conv3p_neighbor_count_f32(const float *points, const int32_t *stride_xyz, float voxel_size,
                              int B, int N, int fz, int fy, int fx, int32_t *count,
                              void *workspace, size_t workspace_bytes, void *stream);
int conv3p_neighbor_count_f64(const double *points, const int32_t *stride_xyz, double voxel_size,
                              int B, int N, int fz, int fy, int fx, int32_t *count,
                              void *workspace, size_t workspace_bytes, void *stream);

/* SELU and its derivative, the activation the reference applies after every conv3p
 * (/root/reference/selu.py:22-26; pointcnn2_acsd.py:49-67).  y may alias x.
 * conv3p_selu_grad: dx = dy * selu'(x), expressed through the forward OUTPUT y
 * (selu' = scale for y > 0 [x >= 0 maps to y >= 0], y + scale*alpha otherwise). */
int conv3p_selu_f32(const float *x, float *y, size_t n, void *stream);
int conv3p_selu_grad_f32(const float *y, const float *dy, float *dx, size_t n, void *stream);
int conv3p_selu_f64(const double *x, double *y, size_t n, void *stream);
int conv3p_selu_grad_f64(const double *y, const double *dy, double *dx, size_t n, void *stream);
/* dx = (dy_a + dy_b) * selu'(y): the gradient join where a layer's activation feeds both the
 * next conv3p and the feature concat (pointcnn2_acsd.py:49-69). */
int conv3p_selu_grad_add_f32(const float *y, const float *dy_a, const float *dy_b, float *dx,
                             size_t n, void *stream);
int conv3p_selu_grad_add_f64(const double *y, const double *dy_a, const double *dy_b, double *dx,
                             size_t n, void *stream);

/* The models' layer: conv3p followed by SELU (pointcnn2_acsd.py:48-67:
 *   net = selu(conv3p(points, net, W, stride, voxel))), with the activation fused into the op's kernels.
 * conv3p_layer_forward:   output = selu(Conv3p(points, input, filter, stride, voxel)).
 * conv3p_layer_backward:  for a layer whose `input` is itself the OUTPUT of a SELU (every layer but the
 *   first): grad_filter as Conv3pGrad; grad_input = (dX + grad_addend) * selu'(input), i.e. the gradient
 *   w.r.t. the ARGUMENT of the SELU that produced `input`, where dX is Conv3pGrad's grad_input and
 *   grad_addend (may be NULL) is the gradient arriving at `input` from its other consumer (the feature
 *   concat, pointcnn2_acsd.py:66).  `grad_out` is the gradient w.r.t. this layer's conv3p output (already
 *   through this layer's own SELU).  Results equal the unfused sequence conv3p_*_cached + conv3p_selu* up
 *   to the rounding of one multiply. */
int conv3p_layer_forward_cached_f32(const float *points, const float *input, const float *filter,
                                    const int32_t *stride_xyz, float voxel_size, int B, int N, int Cin,
                                    int Cout, int fz, int fy, int fx, float *output, void *cache,
                                    size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_layer_forward_cached_f64(const double *points, const double *input, const double *filter,
                                    const int32_t *stride_xyz, double voxel_size, int B, int N, int Cin,
                                    int Cout, int fz, int fy, int fx, double *output, void *cache,
                                    size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_layer_backward_cached_f32(const float *grad_out, const float *points, const float *input,
                                     const float *filter, const int32_t *stride_xyz, float voxel_size,
                                     int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                                     const float *grad_addend, float *grad_input, float *grad_filter,
                                     void *cache, size_t cache_bytes, const conv3p_cache_config *cfg,
                                     void *stream);
int conv3p_layer_backward_cached_f64(const double *grad_out, const double *points, const double *input,
                                     const double *filter, const int32_t *stride_xyz, double voxel_size,
                                     int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                                     const double *grad_addend, double *grad_input, double *grad_filter,
                                     void *cache, size_t cache_bytes, const conv3p_cache_config *cfg,
                                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * The models' conv3p stack as ONE call per pass (not in the reference, which builds it from op calls in Python:
 * pointcnn2_acsd.py:47-68, scene_seg/pointcnn_scene_seg_acsd.py:51-57).
 *
 *   hidden layer l (l = 0 .. n_hidden-1):  act_l = selu(Conv3p(points, act_{l-1}, filter_l, stride_l, voxel)),
 *       act_{-1} = input (in_channels), every hidden layer has `hidden` output channels;
 *   concat = [act_0 | act_1 | ...]  (B, N, n_hidden*hidden)       -- pointcnn2_acsd.py:68
 *   optional head (num_class > 0):  head = selu(Conv3p(points, concat, filter_head, head_stride, voxel))
 *
 * What the single call buys over 2 x (n_hidden + 1) op calls: one host crossing per pass; the activations are
 * written by the layers' epilogues straight into their column block of `concat` and read from there by the next
 * layer and by the backward (no concat copy, no slice copies); the geometry of all layers is built once per
 * `points` (on `side_stream`, if given, concurrently with the first layers).
 * All layers share the filter extents fz,fy,fx; strides[l] = {sx,sy,sz} of hidden layer l, strides[n_hidden] of the
 * head.  filters / grad_filters are HOST arrays of n_hidden (+1) DEVICE pointers, each tensor laid out as Conv3p's
 * filter.  `cache` as for the *_cached entry points, with slots >= number of distinct strides.
 * Shapes outside the register-resident list return CONV3P_ERR_UNSUPPORTED: compose the stack
 * from the per-op entry points then.
 * ------------------------------------------------------------------------------------------- */
#define CONV3P_STACK_MAX_LAYERS 8
typedef struct conv3p_stack_desc {
    int n_hidden;      /* hidden layers (4 in both models)                                   */
    int in_channels;   /* channels of `input`                                                */
    int hidden;        /* output channels of every hidden layer (9)                          */
    int num_class;     /* output channels of the head layer, 0 = no head (classification)    */
    int fz, fy, fx;    /* filter extents of every layer                                      */
    int32_t strides[CONV3P_STACK_MAX_LAYERS + 1][3];
} conv3p_stack_desc;

/* bytes of scratch conv3p_stack_backward_* needs (0 for an invalid description) */
size_t conv3p_stack_scratch_bytes(const conv3p_stack_desc *desc, int elem_bytes, int B, int N);

/* Enqueue the geometry (sort + every layer's neighbour search) of `points` into `cache` on `stream`, ordered after
 * everything already enqueued on `after_stream` (the stream that produces `points`; may be NULL).  A later
 * conv3p_stack_forward_* with the same `points` pointer and cache waits for it instead of searching, so the searches
 * of the NEXT batch can run under the current batch's backward.  The caller must not modify `points` in between. */
int conv3p_stack_prefetch_f32(const conv3p_stack_desc *desc, const float *points, float voxel_size, int B, int N,
                              void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream,
                              void *after_stream);
int conv3p_stack_prefetch_f64(const conv3p_stack_desc *desc, const double *points, double voxel_size, int B, int N,
                              void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream,
                              void *after_stream);

/* concat (B, N, n_hidden*hidden) receives every hidden activation; head_out (B, N, num_class) or NULL. */
int conv3p_stack_forward_f32(const conv3p_stack_desc *desc, const float *points, const float *input,
                             const float *const *filters, float voxel_size, int B, int N, float *concat,
                             float *head_out, void *cache, size_t cache_bytes, const conv3p_cache_config *cfg,
                             void *stream, void *side_stream);
int conv3p_stack_forward_f64(const conv3p_stack_desc *desc, const double *points, const double *input,
                             const double *const *filters, double voxel_size, int B, int N, double *concat,
                             double *head_out, void *cache, size_t cache_bytes, const conv3p_cache_config *cfg,
                             void *stream, void *side_stream);

/* Backward of the same stack, after conv3p_stack_forward_* on the same cache and points.
 *   grad_concat (B, N, n_hidden*hidden): gradient w.r.t. `concat` from its consumer outside the stack (the dense
 *                head of the classification model); NULL = none.
 *   grad_head   (B, N, num_class): gradient w.r.t. the head activation (num_class > 0).  With a head, grad_concat must
 *                be NULL (neither model of the reference feeds the concat to a second consumer):
 *                CONV3P_ERR_UNSUPPORTED otherwise, before anything is launched.
 *   grad_input  (B, N, in_channels); grad_filters[l] like filters[l] (e.g. views of one fused all-reduce buffer). */
int conv3p_stack_backward_f32(const conv3p_stack_desc *desc, const float *points, const float *input,
                              const float *const *filters, float voxel_size, int B, int N, const float *concat,
                              const float *head_out, const float *grad_concat, const float *grad_head,
                              float *grad_input, float *const *grad_filters, void *scratch, size_t scratch_bytes,
                              void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);
int conv3p_stack_backward_f64(const conv3p_stack_desc *desc, const double *points, const double *input,
                              const double *const *filters, double voxel_size, int B, int N, const double *concat,
                              const double *head_out, const double *grad_concat, const double *grad_head,
                              double *grad_input, double *const *grad_filters, void *scratch, size_t scratch_bytes,
                              void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream);

/* The stack-level passes run the hidden layers as ONE launch per pass where the stack, the cache and the device allow
 * it (pointwise_amd/csrc/conv3p_stack_fused.hpp: per-cloud barriers between the layers).  Diagnostics, SYNCHRONISES the
 * device: how many such launches this cache has seen, and the error bits they left (0 = none; 1: a barrier wait gave up,
 * 2: the tiles of a cloud did not share an XCC).  The reference has no counterpart (one op = one launch sequence there,
 * tf_conv3p_atrous.cu:541-642). */
int conv3p_cache_fused_status(void *cache, unsigned *forward_launches, unsigned *backward_launches, unsigned *error_bits);

/* ---------------------------------------------------------------------------------------------
 * The providers' host pre-step on the device.  The reference prepares every batch with
 * per-cloud numpy loops -- rotate_point_cloud + jitter_point_cloud (/root/reference/modelnet_provider.py:23-75) and
 * sort_point_cloud_xyz / sort_point_cloud_xyz2 (/root/reference/util.py:55-109).  Random numbers stay the caller's.
 *
 * conv3p_augment_f32:  out[b,i,:] = (float)( clip(sigma * noise[b,i,:], +-clip) + (double)(float)(p[b,i,:] . R_b) ),
 *   R_b the rotation about the up (y) axis by the angle whose {cos, sin} is cos_sin[b] (device, double[B][2]; NULL =
 *   no rotation); noise (device, double (B,N,3), e.g. standard normal; NULL = no jitter).  out may alias in.
 * conv3p_sort_xyz_order_f32:  order[b][r] = index of the point that comes r-th in cloud b sorted by x, then y,
 *   then z (what util.py:66-68's three argsorts produce), remaining ties by original index.  data rows have
 *   row_floats floats starting with x, y, z.  N <= 8192 (CONV3P_ERR_UNSUPPORTED beyond).
 * conv3p_gather_rows:  dst[b][r] = src[b][order[b][r]] for rows of row_bytes bytes: applies one order to the
 *   points and to every per-point attribute / label array (sort_point_cloud_xyz2).  dst must not alias src.
 * ------------------------------------------------------------------------------------------- */
int conv3p_augment_f32(const float *points_in, const double *cos_sin, const double *noise, double sigma, double clip,
                       int B, int N, float *points_out, void *stream);
int conv3p_sort_xyz_order_f32(const float *data, int B, int N, int row_floats, int32_t *order, void *stream);
int conv3p_gather_rows(const void *src, const int32_t *order, int B, int N, int row_bytes, void *dst, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The dense head of the classification model (/root/reference/pointcnn2_acsd.py:69-75:
 * view (B, N*36) -> fully_connected 512, selu -> dropout_selu -> fully_connected num_class, selu).
 * tf.contrib.layers.fully_connected is y = activation(x . W + b) with W of shape (K, N).
 *
 * conv3p_fc_forward_f32   y (M, N) = act(x (M, K) . W (K, N) + b (N));  act: 0 = identity, 1 = SELU; b may be NULL.
 * conv3p_fc_backward_f32  given y (the forward OUTPUT) and dy = dL/dy:  dW (K, N) = x^T . dz,  db (N) = sum_m dz,
 *                         dx (M, K) = dz . W^T  with dz = dy * act'(y);  dx and db may be NULL.
 * W is streamed exactly once per pass (it is the 151 MB object of the model); products are exact fp32 on the
 * matrix cores; results are bitwise reproducible.  Constraints: N % 8 == 0, N <= 1024, M <= 128
 * (CONV3P_ERR_UNSUPPORTED otherwise).  Scratch from conv3p_fc_workspace_bytes.
 * ------------------------------------------------------------------------------------------- */
size_t conv3p_fc_workspace_bytes(int M, int K, int N);
int conv3p_fc_forward_f32(const float *x, const float *W, const float *b, int M, int K, int N, int act, float *y,
                          void *workspace, size_t workspace_bytes, void *stream);
int conv3p_fc_backward_f32(const float *x, const float *W, const float *y, const float *dy, int M, int K, int N,
                           int act, float *dx, float *dW, float *db, void *workspace, size_t workspace_bytes,
                           void *stream);

/* Kernel-level timing with HIP events recorded on the caller's stream (bench.py uses it
 * to derive the roofline of the dominant kernel).  Off by default; when enabled every
 * kernel launch of this library is bracketed by an event pair.  read() synchronises the
 * recorded events and returns the number of distinct kernel ids; per id: launches and
 * total milliseconds.  name() gives the kernel-id's label. */
int conv3p_profile_enable(int on);
int conv3p_profile_reset(void);
int conv3p_profile_kinds(void);
const char *conv3p_profile_name(int kind);
int conv3p_profile_read(int kind, uint64_t *launches, double *total_ms);

const char *conv3p_status_string(int status);
int conv3p_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CONV3P_H */
