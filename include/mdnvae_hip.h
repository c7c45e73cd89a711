/* C ABI of libmdnvae_hip.so -- the MI355X (gfx950) kernels behind the connectome-VAE
 * module surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless said otherwise;
 *   - `stream` is a hipStream_t (0 = the null stream); every call is asynchronous on it and
 *     never synchronises, allocates or frees;
 *   - scratch memory is passed in by the caller (`workspace`), sized by the matching
 *     *_workspace_bytes() query;
 *   - return value: 0 = launched; < 0 = argument rejected (nothing launched); > 0 = the
 *     hipError_t of a failed launch.  mdnvae_last_error() returns the message of the last
 *     failure on the calling thread.
 *
 * The reference has no FFI layer: its hot path is eager PyTorch inside each training
 * script.  Each entry point cites the reference lines whose arithmetic it replaces.
 */
#ifndef MDNVAE_HIP_H
#define MDNVAE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int mdnvae_version(void);
const char* mdnvae_last_error(void);

/* ---- reparameterisation + KL + reconstruction loss ------------------------------------ */

enum {
    MDNVAE_RECON_POISSON_RATE = 0, /* sum(r - x*log(r+1e-8))   function/VAE_CNN_NN.py:198 */
    MDNVAE_RECON_POISSON_LOG = 1,  /* sum(exp(r) - x*r)        function/MM_Gate.py:180, function/VAE-CNN-GCN.py:202 */
    MDNVAE_RECON_MSE = 2           /* sum((r-x)^2)             function_FC/VAE_CNN_GCN.py:209, function_FC_SC/SC-FC.py:125 */
};

size_t mdnvae_loss_workspace_bytes(void);

/* z = mu + eps*exp(0.5*logvar) over n = B*L elements.  function/VAE_CNN_NN.py:164-167 with the
 * normal draw supplied by the caller. */
int mdnvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream);
/* grad_mu = grad_z ; grad_logvar = grad_z*eps*0.5*exp(0.5*logvar) (autograd of the line above). */
int mdnvae_reparam_bwd(const float* logvar, const float* eps, const float* grad_z, float* grad_mu,
                       float* grad_logvar, long n, void* stream);

/* loss_function(recon_x, x, mu, logvar)  function/VAE_CNN_NN.py:197-200 (and the two other
 * flavours listed at the enum).  recon,x [B,D]; mu,logvar [B,L]; loss_out[3] = {total, recon
 * term, KL}.  If eps is given and z is not NULL, z [B,L] is written in the same pass. */
int mdnvae_vae_loss_fwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                        const float* logvar, long L, const float* eps, float* z, void* workspace,
                        float* loss_out, void* stream);

/* Gradient of the above: grad_recon [B,D] = grad_loss[0] * d(recon term)/d(recon),
 * grad_mu/grad_logvar [B,L] = grad_loss[1] * d(KL)/d(.); grad_loss is a device pointer to two
 * floats (NULL = {1,1}).  If grad_z (and eps) are given, the reparameterisation's
 * backward is folded in: grad_mu += grad_z, grad_logvar += grad_z*eps*0.5*exp(0.5*logvar). */
int mdnvae_vae_loss_bwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                        const float* logvar, long L, const float* eps, const float* grad_z,
                        const float* grad_loss, float* grad_recon, float* grad_mu, float* grad_logvar,
                        void* stream);

/* Loss and its gradient (upstream gradient 1) in one pass over recon and x. */
int mdnvae_vae_loss_fwdbwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                           const float* logvar, long L, float* grad_recon, float* grad_mu,
                           float* grad_logvar, void* workspace, float* loss_out, void* stream);

/* loss_function(recon, x, mu, logvar) and its gradients for recon[b] = sum_r g_br g_br^T (the plain
 * sum of outer products of the multi-channel GCN decoder, function/MM_Gate.py:126-138, 179-182),
 * straight from the node vectors nodes [B,R,MDNVAE_DEC_NP] in one pass over x: neither recon nor
 * its gradient is written to memory.  Upstream gradient 1.  grad_nodes [B,R,MDNVAE_DEC_NP]
 * (columns >= N zero), grad_mu / grad_logvar [B,L] (both or neither), loss_out[3] = {total, recon
 * term, KL}; workspace of mdnvae_outer_loss_workspace_bytes(B) bytes.  R <= 8, N <= MDNVAE_DEC_NP-1.
 * x_symmetric != 0: the caller vouches that every x[b] equals its transpose (e.g. mdnvae_tri_pack
 * has checked the same buffer); the pass then needs no row sums.  On input that is not symmetric
 * the loss is still right but grad_nodes is not. */
size_t mdnvae_outer_loss_workspace_bytes(int B);
int mdnvae_outer_loss_fwdbwd(int mode, const float* nodes, const float* x, int B, int R, int N, const float* mu,
                             const float* logvar, int L, float* grad_nodes, float* grad_mu, float* grad_logvar,
                             int x_symmetric, void* workspace, float* loss_out, void* stream);

/* ---- multi-channel chain decoder ---------------------------------------------------------
 * R channels; channel r maps z [B,L] through NL small dense layers to a node vector g_r [B,N];
 * out[b, i*N+j] = f( sum_r g_r[b,i]*g_r[b,j] ).  Serves
 *   - the five-branch rank-1 decoder, plain or masked:  function/VAE_CNN_NN.py:169-189,
 *     function/VAE-CNN-GCN.py:166-185, function_FC/VAE_CNN_GCN.py:166-191, function_FC/VAE_CNN_NN.py:165-197
 *   - the multi-channel k-NN GCN decoder:  function/MM_Gate.py:84-93, 104-108, 126-138.
 * All layer weights are packed into one zero-padded panel: wt[r][l][k][n] (k < fan-in; row
 * MDNVAE_DEC_KP-1 = bias) and its transpose wn[r][l][n][k]; masks and the k-NN adjacency are
 * applied by the pack calls.  N <= MDNVAE_DEC_NP-1, L <= MDNVAE_DEC_KP-1, R <= 8, NL <= 4. */
#define MDNVAE_DEC_NP 96
#define MDNVAE_DEC_KP 96

enum { MDNVAE_ACT_NONE = 0, MDNVAE_ACT_SIGMOID = 1, MDNVAE_ACT_TANH = 2, MDNVAE_ACT_RELU = 3 };

size_t mdnvae_decoder_panel_floats(int R, int NL);

/* One layer's panel slices (wt, wn point at the [KP][NP] / [NP][KP] slice of layer (r,l)).
 * Linear or GraphCNN layer: w [n_out,K], optional 0/1 mask [n_out,K] (function/VAE-CNN-GCN.py:62-78),
 * optional bias [n_out]. */
int mdnvae_decoder_pack_linear(const float* w, const float* mask, const float* bias, int n_out, int K,
                               float* wt, float* wn, void* stream);
/* GCNLayer: w [N,kn], adjacency adj [N,kn] (int32 node ids), bias [N]  (function/MM_Gate.py:75-93). */
int mdnvae_decoder_pack_gcn(const float* w, const int* adj, const float* bias, int N, int kn, float* wt,
                            float* wn, void* stream);
/* Gradient panel slice dwt [KP][NP] (row KP-1 = bias gradient) back to parameter layout. */
int mdnvae_decoder_unpack_linear_grad(const float* dwt, const float* mask, int n_out, int K, float* dw,
                                      float* db, void* stream);
int mdnvae_decoder_unpack_gcn_grad(const float* dwt, const int* adj, int N, int kn, float* dw, float* db,
                                   void* stream);

/* The same for all count = R*NL layers of a decoder in one launch each way.  HOST arrays of length
 * count, layer (r,l) at r*NL+l: kind (0 = Linear, 1 = GCN layer), device pointers w / bias (or NULL) /
 * aux (Linear: mask or NULL; GCN: adjacency), n_out and k (the weight is [n_out,k]).  wt / wn / dwt
 * are the whole panels; dw / db (or NULL) the per-layer gradient buffers, overwritten or, with
 * accumulate != 0, added to (autograd's `.grad +=`).  count <= 32. */
int mdnvae_decoder_pack_layers(int count, const int* kind, const void* const* w, const void* const* bias,
                               const void* const* aux, const int* n_out, const int* k, float* wt, float* wn,
                               void* stream);
int mdnvae_decoder_unpack_layer_grads(int count, const int* kind, const float* dwt, const void* const* aux,
                                      const int* n_out, const int* k, void* const* dw, void* const* db, int accumulate,
                                      void* stream);

/* acts: activation code of layer l in bits [2l, 2l+1]; epi_act: activation applied to the sum of
 * outer products; icpt_w/icpt_b [N*N] (both or neither): out = out*icpt_w + icpt_b (the GraphCNN
 * intercept with identity mask, function/VAE-CNN-GCN.py:150,184,219-220); do_exp: out = exp(out);
 * drop [R,B,N]: optional multiplier on the final node vectors (dropout masks already scaled);
 * xlat [B,N] = sum of the first xlat_channels node vectors (the reference returns h1+h2+h3+h4).
 * components = 1: out is [B,R,N*N] with out[b,r] = g_r g_r^T, the per-channel matrices that the
 * trait-gated mixture head uses as component means (no epilogue; build-defined, no reference code).
 * nodes [B,R,MDNVAE_DEC_NP] (required): the final node vectors g_r; the backward takes them back.
 * out = NULL (no intercept / exp / dropout): only the node vectors are produced, for a caller that
 * consumes them directly (mdnvae_mdn_outer_fwdbwd).
 * layer_in (optional) [R][NL][B][KP]: every layer's input rows (column KP-1 = 1), kept for a backward
 * pass that then does not recompute the chain (layer_in_saved below). */
int mdnvae_decoder_fwd(const float* z, const float* wt, int B, int N, int L, int R, int NL, int acts,
                       int epi_act, const float* icpt_w, const float* icpt_b, int do_exp, const float* drop,
                       int xlat_channels, int components, float* out, float* xlat, float* nodes, float* layer_in,
                       void* stream);

/* Weight (+ bias row) gradients of all count = R*NL layers of a decoder in panel form:
 * dwt [count][MDNVAE_DEC_KP][MDNVAE_DEC_NP] = layer_in[rl]^T . layer_dpre[rl] with the two
 * [count][B][96] buffers mdnvae_decoder_bwd has filled (the autograd of the F.linear / GCN layers of
 * function/VAE_CNN_NN.py:172-181, function/MM_Gate.py:86-92 with respect to their weights).  Split
 * over the batch, partial panels added in slab order.  workspace:
 * mdnvae_decoder_wgrad_workspace_floats(B, count) floats; all pointers 16-byte aligned. */
size_t mdnvae_decoder_wgrad_workspace_floats(int B, int count);
int mdnvae_decoder_wgrad(const float* layer_in, const float* layer_dpre, int B, int count, float* workspace, float* dwt,
                         void* stream);

/* floats in each of layer_in / layer_dpre for a batch of B */
size_t mdnvae_decoder_bwd_rows_floats(int B, int R, int NL);
/* Backward: grad_out [B,N*N], optional grad_xlat [B,N] -> grad_z [B,L]; per layer the input rows
 * layer_in [R][NL][B][KP] (column KP-1 = 1) and pre-activation gradients layer_dpre [R][NL][B][NP],
 * from which the packed weight gradient of layer (r,l) is layer_in^T . layer_dpre (a library GEMM);
 * with an intercept, grad_times_act [B,N*N] = (grad_out [*out if do_exp]) * act(sum), whose column
 * sums are the intercept-weight gradient.  grad_out is [B,R,N*N] in component mode.  nodes
 * [B,R,MDNVAE_DEC_NP] are the forward's node vectors, node_grad a workspace of the same size.
 * grad_out = NULL (no intercept / exp / dropout): node_grad is an input, the gradient of the node
 * vectors (columns >= N zero), and the walk starts from it.
 * layer_in_saved != 0: layer_in already holds the rows written by mdnvae_decoder_fwd for the same
 * z and weights (used when there is no dropout multiplier; otherwise the chain is recomputed).
 * grad_z_parts (optional, [R,B,L] workspace): lets that path give a workgroup one channel -- a small batch (at
 * most 4096 samples) runs one workgroup per (tile, channel), a large one (8192 samples and more, at most 4
 * layers, no x_latent) keeps the channel's panels in LDS and the gradients in registers; the per-channel parts
 * are added in channel order. */
int mdnvae_decoder_bwd(const float* z, const float* wt, const float* wn, int B, int N, int L, int R, int NL,
                       int acts, int epi_act, const float* icpt_w, const float* icpt_b, int do_exp,
                       const float* drop, int xlat_channels, int components, const float* grad_out, const float* grad_xlat,
                       float* grad_z, float* layer_in, float* layer_dpre, float* grad_times_act, const float* nodes,
                       float* node_grad, int layer_in_saved, float* grad_z_parts, void* stream);

/* The decoder's second stage on its own: out [B,N*N] = act(sum_r g_r g_r^T) from node vectors
 * nodes [B,R,MDNVAE_DEC_NP], and the gradient of the node vectors from grad_out
 * (torch.bmm(merged.unsqueeze(2), merged.unsqueeze(1)), function_fc_new/ResNet.py:180-183, is R = 1). */
int mdnvae_outer_sum_fwd(const float* nodes, int B, int R, int N, int epi_act, float* out, void* stream);
int mdnvae_outer_sum_bwd(const float* grad_out, const float* nodes, int B, int R, int N, int epi_act, float* grad_nodes,
                         void* stream);

/* out_a[e] = sum_b a[b,e], out_c[e] = sum_b c[b,e] in a fixed order (bias / diagonal-weight gradients). */
size_t mdnvae_colsum2_workspace_floats(int B, int D);
int mdnvae_colsum2(const float* a, const float* c, int B, int D, float* workspace, float* out_a, float* out_c,
                   void* stream);

/* y = exp(x), y = a*b: the `torch.exp` after the dense intercept (function/VAE_CNN_NN.py:188) and its backward. */
int mdnvae_exp(const float* x, float* y, long n, void* stream);
int mdnvae_mul(const float* a, const float* b, float* y, long n, void* stream);

/* ---- companions of the library GEMMs for plain dense layers ----------------------------
 * (FNN towers function/MM_Gate.py:157-162; cVAE function_FC_SC/SC-FC.py:95-113; latent heads
 * function/VAE_CNN_NN.py:160-161; dense intercept :187).  The matrix products of the small layers are
 * rocBLAS calls made by the host (large ones take mdnvae_split_gemm below); these kernels do the rest of
 * F.linear + activation. */
/* y[r,c] = act(y[r,c] + bias[c]) in place (bias may be NULL). */
int mdnvae_bias_act(float* y, const float* bias, long rows, int cols, int act, void* stream);
/* y = act(a + b): the residual joins of function_fc_new/ResNet.py:101-107,150-153 */
int mdnvae_add_act(const float* a, const float* b, float* y, long n, int act, void* stream);
/* grad_pre = grad_y * act'(.) expressed through the activation output y. */
int mdnvae_act_bwd(const float* y, const float* grad_y, float* grad_pre, long n, int act, void* stream);
/* Narrow dense layer, at most 8 inputs and 8 outputs (the gate of the trait-gated mixture head, a few
 * traits -> K logits; BUILD-DEFINED layer, the arithmetic is nn.Linear's): y [B,N] = act(x [B,M] .
 * w[N,M]^T + bias) in one kernel, and grad_w [N,M] = grad_pre^T . x with grad_b [N] (or NULL) = column
 * sums of grad_pre in two, the workgroups' partial sums added in index order; accumulate != 0 adds to
 * grad_w / grad_b.  workspace: mdnvae_narrow_linear_workspace_floats() floats. */
int mdnvae_narrow_linear_fwd(const float* x, const float* w, const float* bias, long B, int N, int M, int act, float* y,
                             void* stream);
size_t mdnvae_narrow_linear_workspace_floats(void);
int mdnvae_narrow_linear_wgrad(const float* grad_pre, const float* x, long B, int N, int M, float* workspace,
                               float* grad_w, float* grad_b, int accumulate, void* stream);

/* out[e] = sum_b a[b,e], fixed summation order (bias gradients); accumulate != 0: out[e] += the sum. */
size_t mdnvae_colsum_workspace_floats(int B, int D);
int mdnvae_colsum(const float* a, int B, int D, float* workspace, float* out, int accumulate, void* stream);

/* Weight gradient of F.linear for a small layer over a long batch: grad_w [N,M] = grad_pre^T . x with
 * grad_pre [B,N], x [B,M] (autograd of the tower / head layers, function/MM_Gate.py:157-165,
 * function/VAE_CNN_NN.py:160-161), split over the batch on the fp32 matrix cores and reduced in a
 * fixed order.  N and M multiples of 4; all operands 16-byte aligned.  Meant for N*M up to a few
 * hundred thousand, where a library GEMM has too few output tiles to fill the chip.  accumulate != 0:
 * grad_w += the product. */
size_t mdnvae_linear_wgrad_workspace_floats(int B, int N, int M);
int mdnvae_linear_wgrad(const float* grad_pre, const float* x, int B, int N, int M, float* workspace, float* grad_w,
                        int accumulate, void* stream);
/* The same launch pair also giving the layer's bias gradient grad_b [N] = column sums of grad_pre (autograd of the
 * bias of the same F.linear): every lane sums the values it feeds to the matrix cores, the slab sums ride along in
 * the reduction launch.  accumulate_b != 0: grad_b += the sums.  Same workspace query. */
int mdnvae_linear_wgrad_bias(const float* grad_pre, const float* x, int B, int N, int M, float* workspace, float* grad_w,
                             int accumulate, float* grad_b, int accumulate_b, void* stream);

/* ---- first dense layer on symmetric connectomes --------------------------------------------
 * `self.fc11(x.view(-1, N*N))` (function/MM_Gate.py:157-165; function_FC_SC/SC-FC.py:95-104) sums
 * w[h,i,j] x[b,i,j] over all N*N entries.  For a symmetric x[b] the same number is the product of
 * the N(N+1)/2 upper-triangle entries with w[h,i,j] + w[h,j,i], at half the GEMM work, and the
 * weight gradients of (i,j) and (j,i) are one value.  t(i,j) = i*N - i(i-1)/2 + (j-i), i <= j.
 * N <= 192. */
/* out [B,ld] (ld >= T = N(N+1)/2, columns T..ld-1 are written as zeros: a row stride that is a
 * multiple of 4 keeps the GEMM operands 16-byte aligned) = upper triangle of x [B,N,N]; *asym is
 * OR-ed with 1 when some x[b,i,j] != x[b,j,i] (a NaN counts as different), so that a caller can
 * refuse input the shortcut does not hold for. */
int mdnvae_tri_pack(const float* x, long B, int N, int ld, float* out, int* asym, void* stream);
/* out [H,ld]: out[h,t(i,j)] = w[h,i,j] + w[h,j,i] for i < j, w[h,i,i] on the diagonal, zeros past T. */
int mdnvae_tri_fold(const float* w, long H, int N, int ld, float* out, void* stream);
/* grad_w [H,N*N]: grad_w[h,i*N+j] = g[t(min(i,j),max(i,j))*stride_t + h*stride_h], g being the
 * weight gradient of the folded layer stored as [H,T] (stride_t = 1) or [T,H] (stride_h = 1);
 * accumulate != 0: grad_w += that. */
int mdnvae_tri_unfold(const float* g, long H, int N, long stride_t, long stride_h, float* grad_w, int accumulate,
                      void* stream);

/* ---- the big first layers on the bf16 matrix cores at fp32 accuracy ---------------------------
 * self.fc11(x) / self.fc12(x) with ReLU (function/MM_Gate.py:157-158) and the weight gradients autograd
 * derives for them.  Every fp32 operand is split into three bf16 parts (x = x1 + x2 + x3 exactly) and
 * the product keeps the six part products of relative size >= 2^-16, accumulated in fp32: the terms
 * left out are of the size of one fp32 rounding.
 * Split operands are opaque buffers of mdnvae_split_operand_bytes(rows, k) bytes holding an operand
 * [rows][k] (rows padded to a multiple of 32, k to a multiple of 16) in MFMA-fragment tiles. */
size_t mdnvae_split_operand_bytes(long rows, long k);
/* operand[row_tile0 * 32 + r][c] = src[r][c] for src [R][ld] fp32, c < K (any K <= ld: the operand's k
 * extent is K rounded up to 16, zero filled; rows of any alignment). */
int mdnvae_split_rows(const float* src, long R, long K, long ld, long row_tile0, void* out, void* stream);
/* operand[row_tile0 * 32 + c][r] = src[r][c] for src [Kd][ld] fp32, c < R: the transposed operand (of
 * k extent Kd rounded up to 16, zero filled) that a weight-gradient product takes. */
int mdnvae_split_cols(const float* src, long Kd, long R, long ld, long row_tile0, void* out, void* stream);
/* The same with the ReLU backward of the producing layer folded in (function/MM_Gate.py:157-158: F.relu of
 * fc11 / fc12, whose autograd multiplies the output gradient by output > 0): gate [Kd][ld] or NULL - an
 * entry of src counts only where its gate is positive.  ones_row != 0: operand row R (the one after the
 * last source column) is 1 for k < Kd, so row R of a product with this operand is the column sum of the
 * other one (a bias gradient).  parts (1..3): how many bf16 parts are written (2 for a three-term product). */
int mdnvae_split_cols_gated(const float* src, const float* gate, long Kd, long R, long ld, int ones_row, int parts,
                            long row_tile0, void* out, void* stream);
/* The same operand when the output gradient is itself the data gradient of the next dense layer
 * (function/MM_Gate.py:159-160, fc111 / fc222 on the outputs of fc11 / fc12): operand[row_tile0 * 32 + c][r] =
 * gate[r][c] > 0 ? (gpre . w)[r][c] + dense[r][c] : 0 with gpre [Kd][J], w [J][R] (the next layer's weight as
 * stored), dense [Kd][R] or NULL, gate [Kd][R]; J is 64 or 128, R a multiple of 4.  The product runs on the
 * fp32 matrix cores inside the splitting pass; the [Kd][R] gradient is never written. */
int mdnvae_split_cols_dgrad(const float* gpre, const float* w, const float* dense, const float* gate, long Kd, long R,
                            long J, int parts, long row_tile0, void* out, void* stream);
/* out[m][n] = sum_k a[m][k] b[n][k] (+ bias[n] when biases are given, then ReLU when relu != 0) for m < m_valid: columns
 * n < n_split go to out0 (bias0), the others to out1 (bias1) at column n - n_split; row stride ld_out.
 * M a multiple of tile_m (128 or 256), N and n_split multiples of 256, K of 16.  Only rows < m_valid and, of
 * each output, columns < n_valid are stored; accumulate = 1 adds to what the output holds (no ReLU then);
 * accumulate = 2 does so with the k range split over two workgroups per tile that add atomically (for products
 * with few output tiles and a long k; the output must hold zeros or the value to add to - two addends on top
 * of it, so the result does not depend on their order).
 * terms = 6: all six bf16 products of the three-part operands (error 2^-24 of |a||b|, one fp32 rounding);
 * terms = 3: a1 b1 + a1 b2 + a2 b1 only (error up to 3 * 2^-16 per product, for weight gradients, which
 * nothing downstream amplifies), at half the matrix-core work. */
int mdnvae_split_gemm(const void* a, const void* b, long M, long N, long K, long m_valid, long n_valid,
                      const float* bias0, const float* bias1, int relu, int accumulate, float* out0, float* out1,
                      long n_split, long ld_out, int tile_m, int terms, void* stream);

/* ---- the same products from two fp16 parts per operand ----------------------------------------
 * Again self.fc11 / self.fc12 (function/MM_Gate.py:157-158) and every other large nn.Linear the split product
 * serves (function/VAE_CNN_NN.py:187, function_FC_SC/SC-FC.py:95-113).  x 2^e = h + l with h = fp16(x 2^e),
 * l = fp16(x 2^e - h) carries 22 significand bits, and h_a h_b + h_a l_b + l_a h_b is three fp16 matrix-core
 * products (fp32 accumulation) - half the work of the six bf16 products - with an error of 3 * 2^-22 of
 * sum |a||b| at most (measured: the level of an fp32 dot product).  fp16 has 30 binades, so each TENSOR is
 * scaled by the power of two that brings its largest magnitude into [2^14, 2^15); entries more than 2^17
 * below the tensor's largest lose low-part bits (absolute error below 2^-29 of the largest entry).  The
 * largest magnitude lives on the device as bit patterns of floats: `absmax` is an array of
 * MDNVAE_ABSMAX_SLOTS unsigned partial maxima (the kernels that fill it spread their atomics over the slots,
 * the kernels that use it take the largest). */
#define MDNVAE_ABSMAX_SLOTS 64
/* absmax = max(absmax, |floor_value|, largest |src[r][c]|), src [rows][ld], c < cols; reset != 0 clears the
 * slots first (a one-workgroup kernel on the stream).  Several tensors that share a scale are folded into one array by
 * consecutive calls with reset = 0. */
int mdnvae_absmax(const float* src, long rows, long cols, long ld, float floor_value, int reset, unsigned* absmax,
                  void* stream);
/* mdnvae_tri_pack / mdnvae_tri_fold that also fold the largest magnitude they write into *absmax (cleared first
 * when reset != 0), so the fp16 split of their output needs no pass of its own. */
int mdnvae_tri_pack_absmax(const float* x, long B, int N, int ld, float* out, int* asym, int reset, unsigned* absmax,
                           void* stream);
int mdnvae_tri_fold_absmax(const float* w, long H, int N, int ld, float* out, int reset, unsigned* absmax, void* stream);
/* mdnvae_split_rows / mdnvae_split_cols_gated writing the two fp16 parts of src * 2^e(*absmax); *absmax must
 * be at least the largest magnitude of the source entries written into the operand.  ones_row != 0: operand row R
 * holds 2^14 in its high part whatever the scale (ones times the scale could leave fp16's range); pass R as
 * unit_row to mdnvae_split_gemm_f16 and row R of the product is the column sum of the other operand. */
int mdnvae_split_rows_f16(const float* src, long R, long K, long ld, long row_tile0, const unsigned* absmax, void* out,
                          void* stream);
int mdnvae_split_cols_f16(const float* src, const float* gate, long Kd, long R, long ld, int ones_row, long row_tile0,
                          const unsigned* absmax, void* out, void* stream);
/* Both operands a batch src [Kd][ld] takes part in, from one read of it: out_rows = what mdnvae_split_rows_f16(src,
 * Kd, ld, ld, 0, absmax, ...) writes (the forward product's operand: every column up to ld), out_cols = the
 * weight-gradient product's operand (columns < R and the row of ones): parts = 1..3: that many bf16 parts, as
 * mdnvae_split_cols_gated(src, NULL, Kd, R, ld, ones_row, parts, 0, ...) writes them; parts = 0: two fp16 parts at
 * the same scale, as mdnvae_split_cols_f16(src, NULL, Kd, R, ld, ones_row, 0, absmax, ...) writes them.  ld a
 * multiple of 16, 16-byte aligned rows. */
int mdnvae_split_rows_f16_cols(const float* src, long Kd, long R, long ld, int ones_row, int parts, const unsigned* absmax,
                               void* out_rows, void* out_cols, void* stream);
/* mdnvae_split_cols_dgrad writing two fp16 parts: the gradient of relu(fc11) / relu(fc12) formed from the next
 * layer's (grad_pre, weight) under the gate (function/MM_Gate.py:157-160), scaled by 2^e(*absmax).  *absmax must
 * bound |gpre . w + dense|: mdnvae_dgrad_bound gives such a bound without forming the product.  absmax_gpre (the
 * record of |gpre|, or NULL): with it the product itself runs on the 16-bit matrix cores, both factors as two fp16
 * parts (gpre scaled by its record, each 64-column slice of w by its own largest magnitude; error <= 3 * 2^-22 of
 * sum |gpre||w|); without it, on the fp32 matrix instruction. */
int mdnvae_split_cols_dgrad_f16(const float* gpre, const float* w, const float* dense, const float* gate, long Kd, long R,
                                long J, long row_tile0, const unsigned* absmax, const unsigned* absmax_gpre, void* out,
                                void* stream);
/* absmax = max(absmax, |gpre|max * max_c sum_j |w[j][c]| + |dense|max) for w [J][R]: an upper bound of every entry
 * of gpre . w + dense from the two tensors' largest magnitudes (absmax_gpre, absmax_dense or NULL: arrays
 * mdnvae_absmax filled).  A bound some binades above the true maximum only moves the fp16 split's absolute error
 * floor (2^-29 of the bound); reset != 0 clears the slots first. */
int mdnvae_dgrad_bound(const unsigned* absmax_gpre, const float* w, long J, long R, const unsigned* absmax_dense, int reset,
                       unsigned* absmax, void* stream);
/* mdnvae_split_gemm on two such operands (three products); the epilogue divides by both scales before bias and
 * ReLU.  Arguments as mdnvae_split_gemm; unit_row: the row of `a` written with ones_row != 0 (its products are
 * divided by 2^14 instead of a's scale), or -1. */
int mdnvae_split_gemm_f16(const void* a, const void* b, const unsigned* absmax_a, const unsigned* absmax_b, long M, long N,
                          long K, long m_valid, long n_valid, const float* bias0, const float* bias1, int relu,
                          int accumulate, float* out0, float* out1, long n_split, long ld_out, int tile_m, long unit_row,
                          void* stream);

/* ---- conv / ResNet encoder ---------------------------------------------------------------
 * The ATen conv2d / instance_norm / relu / max_pool2d calls (and their autograd) of the encoder
 * stack, function/VAE_CNN_NN.py:83-98 (ResidualBlock), :110-137 (ec, max_pool), :155-159 (encode).
 * Tensors are NCHW fp32.  Convolutions run on the fp32 matrix cores; the instance norm + ReLU of
 * the producing layer can be applied while a convolution loads its input (in_stats / relu_in),
 * so normalised activations are only materialised where a residual skip needs them. */

/* floats of the packed weight buffers of one conv layer (transposed = the data-gradient copy) */
size_t mdnvae_conv_packed_floats(int Cout, int Cin, int KS, int transposed);
/* w [Cout,Cin,KS,KS] -> packed_fwd ([tap][cin][cout padded]) and, if not NULL, packed_bwd
 * ([flipped tap][cout][cin padded]) used by the data gradient.  Cin == 1: forward copy only. */
int mdnvae_conv_pack_weights(const float* w, int Cout, int Cin, int KS, float* packed_fwd, float* packed_bwd,
                             void* stream);
/* transposed = 0: out [B,Cout,OH,OW] = conv(pad(act(x))) + bias, x [B,Cin,H,W]; padding zero or
 *   reflection (nn.ReflectionPad2d, stride 1 only); in_stats [B*Cin][2] = {mean, rstd} (optional)
 *   normalises x on load and relu_in applies ReLU after it.
 * transposed = 1: data gradient of the convolution with (KS, stride, pad): x is the gradient
 *   w.r.t. the conv output [B,Cin,H,W], packed_w the packed_bwd buffer, out [B,Cout,OH,OW] the
 *   gradient w.r.t. the conv input (for a reflection-padded layer pass pad = 0 and the padded plane
 *   size; fold it with mdnvae_inorm_bwd(folded=1) / mdnvae_fold_add).
 * KS in {3,7}; stride in {1,2}; Cin == 1 (7x7 stem) or a multiple of 8. */
int mdnvae_conv2d(const float* x, const float* in_stats, int relu_in, const float* packed_w, const float* bias,
                  float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                  int reflect, int transposed, void* stream);
/* The 7x7 single-channel stem (function/VAE_CNN_NN.py:83-86: ReflectionPad2d(3) + Conv2d(1, 48, 7)) with the
 * InstanceNorm2d statistics of its output planes taken on the way out: stats [B * Cout][2] = {mean,
 * 1 / sqrt(var + 1e-5)} as mdnvae_plane_stats gives them, without another read of the planes.  Each 256-pixel
 * tile leaves the mean and the sum of squared deviations of its own pixels (workspace), combined per plane in
 * double. */
size_t mdnvae_conv_stem_stats_workspace_floats(int B, int Cout, int OH, int OW);
int mdnvae_conv_stem_stats(const float* x, const float* packed_w, const float* bias, float* out, int B, int H, int W,
                           int Cout, int KS, int pad, float* workspace, float* stats, void* stream);
/* The same convolution / data gradient on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
 * accumulation).  parts = 1: both operands rounded to bf16 - the reduced-precision mode of the encoder of
 * function_FC/VAE_CNN_GCN.py:97-137; results differ from mdnvae_conv2d by the rounding of the inputs (2^-9
 * relative per operand).  parts = 3: each operand is split into three bf16 numbers whose sum is the fp32
 * value and the six part products of relative size >= 2^-16 are accumulated: fp32 accuracy (the same
 * 1e-5 bar as mdnvae_conv2d) at 6/16 of the cost of the fp32 matrix instruction.  parts = 2: three part
 * products, 2^-16 relative per product.  3x3 kernels, Cin and Cout multiples of 16; the 7x7 stem stays
 * with mdnvae_conv2d.  packed_w comes from mdnvae_conv_pack_weights_bf16 with the same parts (whole MFMA
 * operand fragments, [part][tap][16-channel chunk][32-channel tile][lane][8 bf16]);
 * mdnvae_conv_packed_bytes_bf16 is its size in bytes (0 for an unsupported shape). */
size_t mdnvae_conv_packed_bytes_bf16(int Cout, int Cin, int KS, int transposed, int parts);
int mdnvae_conv_pack_weights_bf16(const float* w, int Cout, int Cin, int KS, int parts, void* packed_fwd,
                                  void* packed_bwd, void* stream);
int mdnvae_conv2d_bf16(const float* x, const float* in_stats, int relu_in, const void* packed_w, const float* bias,
                       float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                       int reflect, int transposed, int parts, void* stream);
/* Weight gradient on the bf16 matrix cores with the operands in `parts` bf16 parts (as mdnvae_conv2d_bf16):
 * same arguments, conventions and workspace as mdnvae_conv2d_wgrad; 3x3, pad 1, stride 1 or 2, Cin > 1. */
int mdnvae_conv2d_wgrad_bf16(const float* x, const float* in_stats, int relu_in, const float* grad_out,
                             float* workspace, float* grad_w, int B, int Cin, int H, int W, int Cout, int OH, int OW,
                             int KS, int stride, int pad, int reflect, int parts, void* stream);
/* Weight gradient of the 7x7 single-channel stem (function/VAE_CNN_NN.py:83-86) with both operands rounded to bf16
 * (fp32 accumulation): the bf16 mode's form of mdnvae_conv2d_wgrad(Cin = 1); same workspace, reflection padding.
 * grad16 = 0: grad_out is fp32 [B][Cout][OH * OW]; 1: bf16 planes [B][Cout][pitch], pitch = OH * OW rounded up to a
 * multiple of 8 (as mdnvae_cb_norm_bwd leaves them with omode 3) -- the same result, the rounding having moved into
 * the producer, at half the bytes. */
int mdnvae_conv_stem_wgrad_bf16(const float* x, const void* grad_out, float* workspace, float* grad_w, int B, int H,
                                int W, int Cout, int KS, int pad, int grad16, void* stream);
size_t mdnvae_conv_wgrad_workspace_floats(int B, int Cout, int Cin, int KS);
/* grad_w [Cout,Cin,KS,KS] = sum_{b,pixels} grad_out x pad(act(x)); same x / in_stats / relu_in
 * / padding conventions as the forward call.  3x3 multi-channel or the 7x7 single-channel stem. */
int mdnvae_conv2d_wgrad(const float* x, const float* in_stats, int relu_in, const float* grad_out,
                        float* workspace, float* grad_w, int B, int Cin, int H, int W, int Cout, int OH, int OW,
                        int KS, int stride, int pad, int reflect, void* stream);

/* stats[plane] = {mean, 1/sqrt(biased var + 1e-5)} of each of nplanes planes of P <= 8192 floats. */
int mdnvae_plane_stats(const float* x, long nplanes, int P, float* stats, void* stream);
/* Instance norm in one pass over raw: stats[plane] = {mean, rstd} and
 * out = (skip ? skip : 0) + [relu]((raw - mean)*rstd)  (InstanceNorm2d + ReLU, or x + block(x)). */
int mdnvae_inorm_fwd(const float* raw, const float* skip, int relu, long nplanes, int P, float* out, float* stats,
                     void* stream);
/* out = (skip ? skip : 0) + [relu]((raw - mean)*rstd) from given stats: the materialised activation
 * (x + block(x) of the ResidualBlock when skip is given). */
int mdnvae_norm_act_add(const float* raw, const float* stats, const float* skip, int relu, long nplanes, int P,
                        float* out, void* stream);
/* Backward of [relu](instance_norm(raw)): grad_raw from grad_act.  folded = 1: grad_act lives on
 * the reflection-padded (pad 1) plane [(H+2)x(W+2)] and is folded back while loading.
 * plane_gsum [nplanes] (optional) = per-plane sum of grad_raw (summed over the batch it is the
 * bias gradient of the convolution that produced raw). */
int mdnvae_inorm_bwd(const float* raw, const float* stats, const float* grad_act, int relu, int folded,
                     long nplanes, int H, int W, float* grad_raw, float* plane_gsum, void* stream);
/* out = (base ? base : 0) + fold(grad_padded): gradient reaching a residual block's input. */
int mdnvae_fold_add(const float* base, const float* grad_padded, long nplanes, int H, int W, float* out,
                    void* stream);
/* K x K / stride K max pool over planes [H,W] -> feat [nplanes, (H/K)*(W/K)], argmax = flat
 * position inside the plane; and its backward. */
int mdnvae_maxpool_fwd(const float* x, long nplanes, int H, int W, int K, float* feat, int* argmax, void* stream);
int mdnvae_maxpool_bwd(const float* grad_feat, const int* argmax, long nplanes, int H, int W, int K, float* grad_x,
                       void* stream);

/* ---- conv / ResNet encoder on channel-blocked 16-bit operands --------------------------------
 * The same layers (function/VAE_CNN_NN.py:83-98 ResidualBlock, :110-134 encoder stack; copies at
 * function/VAE-CNN-GCN.py:80-131, function_FC/VAE_CNN_GCN.py:82-137) with every activation that feeds a
 * 3x3 convolution kept as   act[part][group-plane][B][PP positions][8 channels]   (16-bit elements):
 * written once, already padded and rounded, by the normalisation pass that produces it, so that the
 * convolutions copy their operands straight into LDS (no staging arithmetic).  parts = 1: bf16;
 * parts = 2: two fp16 parts of value * scale.  Convolutions write planes out[B][channels][Qr],
 * Qr = rows * pitch rounded up to 8, over the pitched grid of their input (columns past the plane's
 * width are scratch): fp32, or (out16 = 1, forward kinds) fp16 clamped to the finite range - a raw
 * convolution output is normalised and rounded to bf16 by the next pass, which 11 significant bits
 * do not disturb, and a workgroup's write-out is bound by bytes per CU; or (out16 = 2, data-gradient kinds,
 * parts = 1) bf16: the pass that reads a data gradient of the bf16 mode rounds its own result to bf16.
 *
 * kind of product: 0 forward of a stride-1 reflection-pad-1 layer (input: reflected border, pitch W + 2;
 *   rows = H);  1 forward of a stride-2 pad-1 layer (input: four parity planes with a zero block row /
 *   column in front, pitch OW + 1; rows = OH);  2 data gradient of a stride-1 layer (input: output
 *   gradient with a zero border of 2, pitch W + 2; rows = H + 2: the result lives on the reflection-
 *   padded grid and is folded by mdnvae_cb_norm_bwd);  3 data gradient of a stride-2 layer as a 2x2-tap
 *   convolution (input: output gradient with a zero row / column behind, pitch OW + 1; rows = OH; the
 *   result is the four parity planes [B][4 Cin][Qr] of the input gradient).
 * Cout / Cin are those of the LAYER in every kind.  PP: positions per (group-plane, sample) of the
 * input, at least Qr + the largest tap offset. */
size_t mdnvae_cb_packed_bytes(int Cout, int Cin, int kind, int parts);
/* w [Cout][Cin][3][3] -> MFMA B-operand fragments in the order the product of this kind streams them;
 * amax (MDNVAE_ABSMAX_SLOTS partial maxima of |w| as bit patterns) is read for parts = 2 only. */
int mdnvae_cb_pack_weights(const float* w, int Cout, int Cin, int kind, int parts, const unsigned* amax,
                           void* packed, void* stream);
/* The same for n layers (at most 16) in one launch: host arrays of n device pointers / shapes; amax may be NULL
 * for parts = 1. */
int mdnvae_cb_pack_weights_batch(int n, const float* const* w, const int* Cout, const int* Cin, const int* kind, int parts,
                                 const unsigned* const* amax, void* const* packed, void* stream);
/* Two-part operands (parts = 2; fp32-accurate: three fp16 products per multiply-add, error <= 3 * 2^-22 of sum |a||w|):
 * the accumulators are multiplied by out_scale and divided by the power-of-two scales of the records absmax_act (NULL:
 * the input was written with a fixed scale, whose inverse is out_scale) and absmax_w (the record the weights were packed
 * with). */
int mdnvae_cb_conv(const void* act, long part_stride, const void* packed_w, const float* bias, void* out, int out16,
                   int B, int Cout, int Cin, int kind, int rows, int pitch, int PP, int parts, float out_scale,
                   const unsigned* absmax_act, const unsigned* absmax_w, void* stream);
/* stats[plane] = {mean, 1/sqrt(var + 1e-5)} (nn.InstanceNorm2d defaults) over the H x W valid entries of
 * planes stored with a row pitch (planes of at most 2048 positions). */
int mdnvae_cb_plane_stats(const float* y, long nplanes, int H, int W, int pitch, long plane_stride, float* stats,
                          void* stream);
/* a = [relu]((y - mean) * rstd) (+ skip), function/VAE_CNN_NN.py:113-114,97: written as the next
 * convolution's operand (mode 0: reflected border of 1, pitch W + 2, for kind 0; mode 1: four parity
 * planes, pitch (W + 1) / 2 + 1, for kind 1; every one of the PP positions is written) and, when
 * stream_out is given, as fp32 planes [B][C][H * W] (the residual stream).  skip: [B][C][H * W].
 * compute_stats = 1: stats [B * C][2] is an output - the pass stages its planes in LDS and takes
 * {mean, rstd} from them first (planes of a multiple of 4 and at most 5000 positions);
 * compute_stats = 0: stats is read (mdnvae_plane_stats / mdnvae_cb_plane_stats).
 * y16 = 1 (here and in mdnvae_cb_norm_bwd): y holds fp16 planes (mdnvae_cb_conv with out16).
 * parts = 2: the operand is written as two fp16 parts of a * scale (a power of two with |a| * scale < 65504: a
 * normalised plane is bounded by sqrt(H W), each residual block adds as much). */
int mdnvae_cb_norm_fwd(const void* y, int y16, float* stats, int compute_stats, const float* skip, float* stream_out,
                       void* act, long part_stride, int B, int C, int H, int W, int pitch_y, long Qy, int relu, int mode,
                       int PP, int parts, float scale, void* stream);
/* Backward of the same pass, fused with what comes before it: the gradient w.r.t. the activation is
 * gathered from the consumer's data gradient (gmode 0: planes on this grid with pitch_g; 1: the
 * reflection-padded grid of kind 2, folded back; 2: the parity planes of kind 3; 3: none), plus `base`
 * ([B][C][H * W], the residual stream's gradient) if given; the sum is written to sum_out if given;
 * then the ReLU mask and the instance-norm backward
 *     dc = rstd * (dn - mean(dn) - xh * mean(dn * xh)),  xh = (y - mean) * rstd
 * and the result leaves as the operand of the layer's own data / weight gradient (omode 0: zero
 * border of 2, pitch W + 2, H + 4 rows, for kind 2;  omode 1: zero row / column behind, pitch W + 1,
 * for kind 3), as fp32 planes [B][C][H * W] (omode 2), or as bf16 planes [B][C][PP] with PP >= H * W a multiple
 * of 8 (omode 3, with g16 = 2: what mdnvae_conv_stem_wgrad_bf16 reads with grad16 = 1).  gsum [B * C]: plane sums
 * of dc (the convolution-bias gradient before the sum over the batch; summed before any rounding).
 * parts = 2: dc leaves as two fp16 parts scaled by the power of two of the record absmax_dc, the largest |dc|:
 * run the pass once with absmax_probe (a cleared record; nothing else is written) and again with that record as
 * absmax_dc; the same record goes to the products that read dc.  (Any record between the largest |dc| and some
 * 2^6 times it serves: the parts are floating point, and a loose scale only raises the absolute error floor,
 * 2^-40 of the record.) */
int mdnvae_cb_norm_bwd(const void* g, int gmode, int pitch_g, long Qg, const float* base, float* sum_out,
                       const void* y, int y16, const float* stats, int pitch_y, long Qy, int relu, int B, int C, int H,
                       int W, int omode, void* dc, long part_stride, int PP, int parts, float scale, float* gsum,
                       const unsigned* absmax_dc, unsigned* absmax_probe, int g16, void* stream);
/* dW [Cout][Cin][3][3] from the channel-blocked output gradient g (as mdnvae_cb_norm_bwd leaves it) and the
 * layer's channel-blocked input.  planes = 1: stride-1 reflection layer (g: omode 0, act: mode 0, same
 * pitch); planes = 4: stride-2 layer (g: omode 1, act: mode 1).  [kbeg, kend): the positions of g that
 * hold output pixels, widened to multiples of 32 (everything else in g is zero).
 * parts = 2: both operands hold two fp16 parts (part strides in bytes); the result is multiplied by out_scale (the
 * inverse of the activations' fixed scale) and divided by the scale of absmax_g. */
size_t mdnvae_cb_wgrad_workspace_floats(int B, int Cout, int Cin, int planes, int parts);
int mdnvae_cb_wgrad(const void* g, const void* act, float* workspace, float* grad_w, int B, int Cout, int Cin,
                    int planes, int PPg, int PPa, int kbeg, int kend, int pitch, int parts, float out_scale,
                    long g_part_stride, long act_part_stride, const unsigned* absmax_g, void* stream);

/* ---- optimiser ---------------------------------------------------------------------------
 * torch.optim.Adam(lr=1e-3) of function/VAE_CNN_NN.py:204,219 over one flat buffer of n floats
 * (all parameters / gradients / moments concatenated); step counts from 1; grad_scale multiplies
 * the gradient first (1 for a SUM-reduced data-parallel gradient). */
int mdnvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int step, float lr,
                     float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same update with the step count on the device (*step_dev is incremented first, so it starts
 * at 0; coef_dev: two floats of workspace): nothing in the call changes from step to step, which is
 * what a step captured once into a HIP graph and replayed needs. */
int mdnvae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int* step_dev,
                         float* coef_dev, float lr, float beta1, float beta2, float eps, float grad_scale,
                         void* stream);

/* Adam for the weight [H][N*N] of a symmetric-input first layer (function/MM_Gate.py:157-158 on
 * symmetric connectomes) from its gradient folded onto the upper triangle: grad_t [t][H], t the
 * row-major walk of the triangle (what the weight-gradient GEMM on the packed input produces;
 * w_ij and w_ji share the value).  exp_avg_t / exp_avg_sq_t [N(N+1)/2][H] hold the moments in the same
 * layout; both mirror entries of param take the step.  Same numbers as mdnvae_tri_unfold followed by
 * mdnvae_adam_step.  grad_t is consumed (its first N(N+1)/2 rows hold the steps afterwards).
 * coef_dev (or NULL): the device pair a preceding mdnvae_adam_step_dev on the same stream has
 * left; `step` is ignored then. */
int mdnvae_adam_step_folded(float* param, float* grad_t, float* exp_avg_t, float* exp_avg_sq_t, int H, int N,
                            int step, const float* coef_dev, float lr, float beta1, float beta2, float eps,
                            float grad_scale, void* stream);

/* ---- trait-gated Gaussian-mixture likelihood (BUILD-DEFINED: the reference has no such code) ----
 * comp_mean [B,K,D] component means, x [B,D], gate_logits [B,K] (from the traits), log_sigma [K];
 *   nll = -sum_b logsumexp_k( log softmax(gate_logits)_bk
 *                             + sum_d(-0.5((x_bd-m_bkd)/s_k)^2 - log s_k - 0.5 log 2pi) )
 * loss_out[3] = {nll + KL, nll, KL} with KL as in function/MM_Gate.py:181; z [B,L] is written when
 * eps and z are given; resp [B,K] (responsibilities) and sumsq [B,K] are saved for the backward. */
size_t mdnvae_mdn_workspace_bytes(int B);
int mdnvae_mdn_loss_fwd(const float* comp_mean, const float* x, const float* gate_logits, const float* log_sigma,
                        int B, int K, int D, const float* mu, const float* logvar, int L, const float* eps, float* z,
                        float* resp, float* sumsq, void* workspace, float* loss_out, void* stream);
/* grad_comp [B,K,D], grad_logits [B,K], grad_log_sigma_rows [B,K] (sum over b = gradient of log_sigma),
 * grad_mu/grad_logvar [B,L] (KL part; both or neither); grad_loss = device {g_nll, g_kl} or NULL;
 * workspace of mdnvae_mdn_workspace_bytes(B) bytes; B*K*D < 2^32. */
/* The same likelihood for rank-one component means m_bk = g_bk g_bk^T given by their node vectors
 * nodes [B,K,MDNVAE_DEC_NP] (the GCN decoder's channels; N*N = D), forward and backward in one pass
 * over x: the component matrices and their gradients are never written to memory.  Upstream
 * gradient 1.  grad_nodes [B,K,MDNVAE_DEC_NP] (columns >= N zero), grad_logits / resp /
 * grad_log_sigma_rows [B,K], grad_mu / grad_logvar [B,L] (both or neither), loss_out[3] as above;
 * workspace of mdnvae_mdn_workspace_bytes(B) bytes.  K <= 8, N <= MDNVAE_DEC_NP-1.
 * x_symmetric != 0: the caller vouches that every x[b] equals its transpose (e.g. mdnvae_tri_pack
 * has checked the same buffer); the pass then needs no row sums and runs at streaming speed.  On
 * input that is not symmetric the loss is still right but grad_nodes is not.
 * loss_out = NULL: the per-sample terms stay in the workspace (two doubles per sample) and
 * mdnvae_mdn_finalize adds them up later - for a caller that sums several chunks, or times the pass on its own. */
int mdnvae_mdn_outer_fwdbwd(const float* nodes, const float* x, const float* gate_logits, const float* log_sigma,
                            int B, int K, int N, const float* mu, const float* logvar, int L, float* grad_nodes,
                            float* grad_logits, float* grad_log_sigma_rows, float* grad_mu, float* grad_logvar,
                            float* resp, int x_symmetric, void* workspace, float* loss_out, void* stream);
/* loss_out[3] = {nll + KL, nll, KL} from the per-sample terms mdnvae_mdn_outer_fwdbwd (or mdnvae_mdn_loss_fwd) left
 * in `workspace` for B samples: one workgroup, fixed summation order.  grad_log_sigma_rows [B,K] and grad_log_sigma
 * [K] (both or neither): the same launch adds up the per-sample log-sigma gradients, one more workgroup per column. */
int mdnvae_mdn_finalize(const void* workspace, int B, float* loss_out, const float* grad_log_sigma_rows, int K,
                        float* grad_log_sigma, void* stream);
/* mdnvae_mdn_outer_fwdbwd without the workspace argument, for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  The per-sample partial sums live in a static 1 MiB device array
 * inside the library (one copy per GPU, held until the process ends); the call allocates and frees
 * nothing.  B <= 65536, larger batches are rejected (code -2).  Same results as
 * mdnvae_mdn_outer_fwdbwd.  All calls on one GPU share that array: calls that overlap in time on one
 * GPU (two streams, two threads) race; calls on one stream do not. */
int mdnvae_mdn_outer_fwdbwd_static(const float* nodes, const float* x, const float* gate_logits,
                                   const float* log_sigma, int B, int K, int N, const float* mu, const float* logvar,
                                   int L, float* grad_nodes, float* grad_logits, float* grad_log_sigma_rows,
                                   float* grad_mu, float* grad_logvar, float* resp, int x_symmetric, float* loss_out,
                                   void* stream);
int mdnvae_mdn_loss_bwd(const float* comp_mean, const float* x, const float* gate_logits, const float* log_sigma,
                        const float* resp, const float* sumsq, int B, int K, int D, const float* mu,
                        const float* logvar, int L, const float* grad_loss, float* grad_comp, float* grad_logits,
                        float* grad_log_sigma_rows, float* grad_mu, float* grad_logvar, void* workspace,
                        void* stream);

/* ---- FID validation statistics ------------------------------------------------------------
 * np.mean / np.cov(rowvar=False) / sqrtm / trace of calculate_fid (function/VAE_CNN_NN.py:310-316,
 * 331-337; validation/FID.py:61-81).  Feature rows x [n,d] fp32; `shift` [d] is any fixed vector
 * near the mean (the same on every rank); sum [d] and gram [d,d] are float64 accumulators
 * (zeroed by the caller) that add up across calls and, for data-parallel runs, across ranks. */
size_t mdnvae_fid_gram_workspace_floats(long n, int d);
/* sum += sum_rows (x - shift);  gram += sum_rows (x - shift)(x - shift)^T  (MFMA GEMM).  For
 * d > 128 only the 128 x 128 tiles of gram on or above the diagonal are maintained (the matrix is
 * symmetric); mdnvae_fid_finalize reads the rest from their mirror image. */
int mdnvae_fid_accumulate(const float* x, const float* shift, long n, int d, float* workspace, double* sum,
                          double* gram, void* stream);
/* mean [d] = shift + sum/n;  cov [d,d] = (gram - sum sum^T/n)/(n-1), float64. */
int mdnvae_fid_finalize(const double* gram, const double* sum, const float* shift, double n, int d, double* mean,
                        double* cov, void* stream);
/* Elementwise steps of the coupled Newton-Schulz square root (float64; the products are
 * mdnvae_dgemm_sym below): t = 1.5 I - 0.5 zy;  out = scale * (a + a^T)/2;
 * out2 = {trace(a), ||a||_F^2} (above 256 rows summed with float64 atomics: the last bits depend on the
 * order of arrival). */
int mdnvae_ns_residual(const double* zy, int d, double* t, void* stream);
/* The iteration's products on the float64 matrix instruction (in place of the torch.matmul / scipy sqrtm products
 * of function/VAE_CNN_NN.py:310-316): c = a b for d x d float64 matrices with b symmetric (read along its rows), d a
 * multiple of 16.  upper = 1, for a product that is symmetric by its structure (a b = x s x with s symmetric, say):
 * tiles on or above the diagonal are computed only and every value is written to (i, j) and (j, i) -- half the
 * work, and a result symmetric to the bit.  (Not for products that are symmetric only because the factors commute:
 * rounding breaks the commutation, and the mirrored copy keeps the commutator where an average would cancel it.)
 * epilogue 1: c = 1.5 I - 0.5 a b.  c_transposed (upper = 0; may be NULL): the transpose of c, written as well.
 * skip (may be NULL): a device word; the call does nothing if it is non-zero when the kernel starts (state + 4 of
 * mdnvae_ns_state_step: products queued behind the end of the iteration). */
int mdnvae_dgemm_sym(const double* a, const double* b, int d, int upper, int epilogue, double* c, double* c_transposed,
                     const double* skip, void* stream);
/* The iteration's stopping rule on the device (the reference's sqrtm decides inside scipy; here the host queues
 * several iterations and reads the state once per batch): called once per iteration with the new Y.  state [8]
 * (zeroed before the first iteration): [0] previous trace, [1] previous change of the trace, [2] / [3] whether they
 * exist, [4] 0 running / 1 settled / 2 diverged (a non-finite trace or gap), [5] iterations counted, [6] the last
 * trace.  Settled: the trace moved by at most 1e-15 relative, or by less than 1e-9 relative and no less than the
 * step before; or gap (may be NULL; gap[0] = ||I - Z Y||_F^2 from mdnvae_ns_residual_gap) is at most 1e-12.
 * Calls after the end leave the state as it is. */
int mdnvae_ns_state_step(const double* y, int d, const double* gap, double* state, void* stream);
int mdnvae_symmetrize(const double* a, int d, double scale, double* out, void* stream);
int mdnvae_trace_norm(const double* a, int d, double* out2, void* stream);
/* out[0] = max_i sum_j |a_ij|: with the Frobenius norm, the two bounds of the largest eigenvalue the
 * iteration scales by (it needs the spectrum inside (0, 1]; the tighter bound saves iterations). */
int mdnvae_inf_norm(const double* a, int d, double* out, void* stream);
/* y = a x (float64, d x d): the matrix-vector product of a few power-iteration steps, whose Rayleigh quotient sits
 * just under the largest eigenvalue -- a far tighter scale for the iteration than either norm on covariances of
 * many features (each factor 2.25 of looseness costs one iteration); in place of the scaling inside scipy's sqrtm at
 * function/VAE_CNN_NN.py:313. */
int mdnvae_matvec_f64(const double* a, const double* x, int d, double* y, void* stream);
/* The step of the iteration that yields a square-root FACTOR (Y with Y Y^T = A; no counterpart in the
 * reference, which calls scipy's sqrtm at function/VAE_CNN_NN.py:313): w = Z Y is symmetric up to rounding,
 * t = 1.5 I - 0.25 (w + w^T), and gap[0] = ||I - w||_F^2 (cleared by the call, summed with float64
 * atomics, so its last bits depend on the order of arrival).  mdnvae_trace: out[0] = trace(a). */
int mdnvae_ns_residual_gap(const double* w, int d, double* t, double* gap, void* stream);
int mdnvae_trace(const double* a, int d, double* out, void* stream);

/* The same steps in fp32 on padded matrices [d][ld] (ld a multiple of 32; columns d..ld-1 of every output
 * are zero), for the part of the iteration that runs as split-bf16 products (mdnvae_split_gemm) before a
 * few float64 steps finish it: t = 1.5 I - 0.5 zy;  out = scale (a + a^T)/2;  out[0] = trace(a) in
 * float64;  and the conversions between a float64 [d][d] matrix and its padded fp32 copy. */
int mdnvae_ns_residual_f32(const float* zy, int d, long ld, float* t, void* stream);
int mdnvae_symmetrize_f32(const float* a, int d, long ld, float scale, float* out, void* stream);
int mdnvae_trace_f32(const float* a, int d, long ld, double* out, void* stream);
int mdnvae_f64_to_f32_padded(const double* a, int d, long ld, float* out, void* stream);
int mdnvae_f32_padded_to_f64(const float* a, int d, long ld, double* out, void* stream);

/* ---- evaluation loop ------------------------------------------------------------------------
 * function/VAE_CNN_NN.py:268-271: max_value = real.max(); real/max_value; max(generated/max_value, 0);
 * :273-289: per-matrix mean squared error behind compute_mse / compute_psnr. */
size_t mdnvae_max_workspace_floats(void);
/* out[0] = max(x[0..n)) with np.max semantics: a NaN anywhere gives NaN.  x 4-byte aligned. */
int mdnvae_max(const float* x, long n, float* workspace, float* out, void* stream);
/* out = x / denom[0] (denom on the device), clamped below at `lo` when clamp != 0. */
int mdnvae_div_clamp(const float* x, const float* denom, int clamp, float lo, long n, float* out, void* stream);
/* out[r] = mean over cols of (a[r][c]-b[r][c])^2. */
int mdnvae_row_mse(const float* a, const float* b, long rows, int cols, float* out, void* stream);

/* ---- connectome preparation ---------------------------------------------------------------------
 * Per subject: m = raw + raw^T, diagonal := column means, crop [crop_lo, crop_hi)^2, scale, flatten.
 * function/VAE_CNN_NN.py:32-39 (DIVIDE by offset), function_FC/VAE_CNN_GCN.py:35-43 (UNIT: (x+1)/2),
 * function/Full_NN.py:106-112 (LOG1P), function_fc_new/ResNet.py:17-19,33-41 (FISHER z);
 * RAW: crop only (the group-mean matrix of function/VAE_CNN_NN.py:21 averages raw crops).
 * raw [B,n,n] -> out [B,(crop_hi-crop_lo)^2]. */
enum { MDNVAE_PREP_DIVIDE = 0, MDNVAE_PREP_UNIT = 1, MDNVAE_PREP_LOG1P = 2, MDNVAE_PREP_FISHER = 3, MDNVAE_PREP_RAW = 4 };
int mdnvae_prep_connectome(const float* raw, long B, int n, int crop_lo, int crop_hi, int mode, float offset,
                           float* out, void* stream);

/* ---- batched evaluation metrics -----------------------------------------------------------------
 * out [count,k] = the k largest eigenvalues, descending, of each symmetric n x n matrix (n <= 96):
 * compute_spectral_features, function/VAE_CNN_NN.py:302-308, for symmetric input (the matrix is
 * symmetrised as (A + A^T)/2 on load). */
int mdnvae_spectrum_topk(const float* mats, long count, int n, int k, float* out, void* stream);
/* out [count] = mean SSIM of each image pair: 7x7 uniform window, sample covariance, K1 0.01, K2 0.03
 * (the skimage defaults behind compute_ssim_average, function/VAE_CNN_NN.py:291-297). */
int mdnvae_ssim(const float* a, const float* b, long count, int n, float data_range, float* out, void* stream);
/* out [count,5] = {weighted-degree mean, std, weighted clustering mean, std, average shortest path
 * length} of each n x n matrix read as a weighted graph (n <= 128): compute_graph_features of
 * validation/FID.py:6-58 (weighted_paths = 1: edge distance 1/(w + 1e-5)) and of
 * function/VAE_CNN_NN.py:341-379 (weighted_paths = 0: hop counts, as that script attaches no distance). */
int mdnvae_graph_features(const float* mats, long count, int n, int weighted_paths, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDNVAE_HIP_H */
