/*
 * conv3p CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A from-scratch plain-C restatement of the reference CPU operator
 *   /root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp   (Conv3p / Conv3pGrad, CPUDevice)
 * used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker.  Nothing under pointwise_amd/ may link, import or call it.
 *
 * Pinning status (see oracle/README.md and DESIGN.md "Oracle"):
 *   - neighbour search / binning / per-bin counts (reference Grid, .cpp:138-388):
 *     PINNED against the reference's own Grid template, compiled in place from the
 *     TF-free line range of the reference file (oracle/Makefile -> oracle/_ref/).
 *   - the Compute() accumulation loops (.cpp:451-504, :608-716): PINNED bit-for-bit
 *     (f32 and f64, serial) against the reference's own loop text, compiled in place
 *     between the parts of oracle/ref_compute_driver.cpp, which supplies the locals
 *     those lines read (sizes, pointers, `*_flat` views) in place of the TensorFlow
 *     tensors Compute() unpacks them from (oracle/Makefile -> oracle/_ref/libref_compute_*;
 *     tests/test_oracle.py).  The Compute() prologues (OP_REQUIRES, allocate_output) are
 *     TensorFlow-bound and are not executed; the golden fixtures hold the reference
 *     loops' outputs.
 *
 * All entry points return 0 on success, a negative code on invalid arguments.
 * Layouts (row-major, identical to the reference op):
 *   points (B,N,3)  input (B,N,Cin)  filter (fz,fy,fx,Cin,Cout)  output (B,N,Cout)
 *   stride_xyz = {sx,sy,sz}  (reference order, .cpp:438-440)
 *   filter tap index f = (bz*fy + by)*fx + bx                     (.cpp:290)
 *   weight index     (f*Cin + k)*Cout + c                          (.cpp:490)
 */
#ifndef CONV3P_ORACLE_H
#define CONV3P_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define CONV3P_ORACLE_OK 0
#define CONV3P_ORACLE_EINVAL (-1)

/* nthreads <= 1 : serial semantics (reference build without CONV_OPENMP) -- the
 *                 deterministic variant every parity test uses.
 * nthreads  > 1 : OpenMP over the batch dimension with per-thread grad_filter
 *                 partials summed in thread order (reference build with
 *                 -DCONV_OPENMP, .cpp:608-622, :709-716) -- the CPU baseline. */

int conv3p_oracle_forward_f32(const float *points, const float *input, const float *filter,
                              const int *stride_xyz, float voxel_size,
                              int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                              float *output, int nthreads);
int conv3p_oracle_forward_f64(const double *points, const double *input, const double *filter,
                              const int *stride_xyz, double voxel_size,
                              int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                              double *output, int nthreads);

int conv3p_oracle_backward_f32(const float *grad_out, const float *points, const float *input,
                               const float *filter, const int *stride_xyz, float voxel_size,
                               int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                               float *grad_input, float *grad_filter, int nthreads);
int conv3p_oracle_backward_f64(const double *grad_out, const double *points, const double *input,
                               const double *filter, const int *stride_xyz, double voxel_size,
                               int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                               double *grad_input, double *grad_filter, int nthreads);

/* count[(b*N + i)*F + f] = number of accepted neighbours of point i in tap f
 * (reference Grid::neighbor_count, .cpp:306-379). */
int conv3p_oracle_neighbor_count_f32(const float *points, const int *stride_xyz, float voxel_size,
                                     int B, int N, int fz, int fy, int fx, int *count);
int conv3p_oracle_neighbor_count_f64(const double *points, const int *stride_xyz, double voxel_size,
                                     int B, int N, int fz, int fy, int fx, int *count);

/* CSR dump of the accepted-neighbour lists of ONE cloud, in the reference's
 * visit order (cell order oz,oy,ox ascending, original index order inside a
 * cell; .cpp:260-297).  offsets has N+1 entries; nbr_index / nbr_tap need
 * `capacity` entries.  Returns the total number of pairs, or a negative code
 * (-2: capacity too small). */
long conv3p_oracle_neighbor_lists_f32(const float *points, int N, const int *stride_xyz,
                                      float voxel_size, int fz, int fy, int fx,
                                      long *offsets, int *nbr_index, int *nbr_tap, long capacity);
long conv3p_oracle_neighbor_lists_f64(const double *points, int N, const int *stride_xyz,
                                      double voxel_size, int fz, int fy, int fx,
                                      long *offsets, int *nbr_index, int *nbr_tap, long capacity);

/* Backward pair dump of ONE cloud (the set Conv3pGrad actually accumulates,
 * .cpp:647-700): for every centre j, every (ii, tap f', count) that passes the
 * j-centred search, the ii-centred hole test and the count!=0 test. */
long conv3p_oracle_backward_pairs_f32(const float *points, int N, const int *stride_xyz,
                                      float voxel_size, int fz, int fy, int fx,
                                      int *pair_j, int *pair_ii, int *pair_tap, int *pair_count,
                                      long capacity);
long conv3p_oracle_backward_pairs_f64(const double *points, int N, const int *stride_xyz,
                                      double voxel_size, int fz, int fy, int fx,
                                      int *pair_j, int *pair_ii, int *pair_tap, int *pair_count,
                                      long capacity);

int conv3p_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
