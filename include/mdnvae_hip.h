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

#ifdef __cplusplus
}
#endif
#endif /* MDNVAE_HIP_H */
