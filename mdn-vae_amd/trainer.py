"""Data-parallel training step: zero_grad -> forward -> loss -> backward -> SUM all-reduce of
one flat gradient buffer -> Adam, mirroring train_epoch of the reference
(function/VAE_CNN_NN.py:209-222, function/MM_Gate.py:195-208).

One process per GPU.  Parameters, gradients and Adam moments each live in one flat fp32
buffer; the parameters of the module are views into it, so a step issues one RCCL all-reduce
(torch.distributed backend "nccl") and one Adam launch.  The reduction is a SUM, not a mean:
the reference loss is sum-reduced, so G ranks with local batch b take exactly the step a
single process would take on the G*b batch.
"""
import torch
import torch.distributed as dist

from . import ops
from ._lib import call, ptr, stream


class FlatBuffers:
    """Re-homes a module's parameters (and their .grad) into two flat fp32 buffers.  Every
    parameter starts at a multiple of 4 floats, so each view is 16-byte aligned for the kernels
    that write gradients into it; the few padding elements stay zero in both buffers (a zero
    gradient leaves a zero parameter at zero under Adam).  `front(param) -> bool` moves the
    parameters it selects to the start of the buffers (module order otherwise), so that the rest
    is one contiguous range."""

    def __init__(self, module, front=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        if front is not None:
            self.params = [p for p in self.params if front(p)] + [p for p in self.params if not front(p)]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + (s + 3) // 4 * 4)
        total = self.offsets[-1]
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)

    @property
    def numel(self):
        return self.offsets[-1]

    def zero_grad(self, start=0):
        """Zero the gradient buffer from element `start` on (the front may hold gradients that the
        step overwrites or never reads)."""
        self.flat_g[start:].zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach views if something reset .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)


# A one-rank process group normally counts as "not distributed".  Set to 1 to send its gradients
# through the collective calls anyway (the sums are then identities): the only way to run the
# RCCL code path on a machine with a single GPU.
MIN_RANKS_FOR_EXCHANGE = 2


class _Done:
    """A finished collective (what the host-staged path returns where RCCL returns a Work)."""

    def wait(self):
        return True


_HOST_STAGE = {}


def _host_staged(group):
    """gloo moves host memory.  Its own handling of device tensors is not dependable in this image
    (two ranks on one GPU: an all-reduce left pending while others were issued and waited for first
    never completed; an earlier run aborted inside the call), so for a gloo group device buffers are
    copied to pinned host memory, summed there and copied back.  RCCL ("nccl") takes device memory."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum_(t, group=None, async_op=False):
    """In-place SUM all-reduce of one contiguous tensor; returns an object with .wait()."""
    if t.is_cuda and _host_staged(group):
        n = t.numel()
        host = _HOST_STAGE.get(t.dtype)
        if host is None or host.numel() < n:
            host = _HOST_STAGE[t.dtype] = torch.empty(max(n, 1 << 20), dtype=t.dtype, pin_memory=True)
        view = host[:n].view(t.shape)
        view.copy_(t)                                  # blocking: the producing kernels have finished
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        t.copy_(view)
        return _Done()
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else _Done()


def broadcast_(t, group=None):
    """In-place broadcast of one contiguous tensor from the group's first rank."""
    src = dist.get_global_rank(group, 0) if group is not None else 0
    if t.is_cuda and _host_staged(group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
        return
    dist.broadcast(t, src=src, group=group)


def allreduce_sum(flat, group=None, bucket_floats=8 << 20):
    """In-place SUM all-reduce of a flat buffer in buckets of `bucket_floats` (32 MiB): large
    enough that each xGMI ring step moves megabytes, small enough to pipeline."""
    if not _distributed(group):
        return
    works = []
    for o in range(0, flat.numel(), bucket_floats):
        works.append(all_reduce_sum_(flat[o:o + bucket_floats], group, async_op=True))
    for w in works:
        w.wait()


def _same_buffer(a, b):
    return isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.data_ptr() == b.data_ptr() \
        and a.numel() == b.numel() and a.is_contiguous() and b.is_contiguous()


def _distributed(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= MIN_RANKS_FOR_EXCHANGE


class Trainer:
    """`overlap_floats`: with more than one rank, a parameter of at least this many elements has
    its gradient all-reduced from a post-accumulate hook, i.e. as soon as backward has produced it,
    so the transfer of the large encoder matrices runs under the rest of the backward pass (every
    rank runs the same graph, so the hooks fire in the same order everywhere).  A symmetric-input
    first layer (GCNVAE(symmetric_input=True)) hands over its weight gradient folded onto the upper
    triangle, so half the bytes cross the links and the copy to both triangles happens after the
    sum.  The rest of the flat gradient buffer is reduced in buckets after backward.

    With more than one rank the parameters of the group's first rank are broadcast at construction, as
    DistributedDataParallel does: replicas start equal whatever each process seeded its initialisation with.

    `fast_wgrad`: the weight-gradient products of the large dense layers keep two bf16 parts per operand
    (ops.arithmetic(wgrad="bf16x3"): about 1e-5 of a gradient's norm, 1e-3 of single entries) instead of the
    default two fp16 parts (fp32 dot-product level); a few percent faster."""

    def __init__(self, model, loss_mode, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, fused_loss=True,
                 overlap_floats=1 << 20, direct_grads=True, folded_adam=True, fast_wgrad=False):
        self.model, self.loss_mode, self.group = model, loss_mode, group
        self.lr, self.betas, self.eps = lr, betas, eps
        self.wgrad_arithmetic = "bf16x3" if fast_wgrad else None
        self._split_graph = False   # a captured data-parallel step: gradients are exchanged between two graphs
        early = _distributed(group) and overlap_floats
        # `folded_adam`: weights of symmetric-input first layers keep their gradient folded onto the
        # upper triangle all the way into Adam (moments in the folded layout, both mirror entries
        # take the step): step() then never writes, zeroes or re-reads their unfolded gradient.
        provider = getattr(model, "folded_grad_params", None)
        self._fold = {id(p): (p, n) for p, n in (provider() if (provider and folded_adam and direct_grads) else [])
                      if p.requires_grad}
        # those weights first, then the other early-reduced matrices: what is left is one range
        self.buf = FlatBuffers(model, front=(lambda p: id(p) in self._fold or bool(early and p.numel() >= overlap_floats))
                               if (early or self._fold) else None)
        if self._fold:
            head = self.buf.params[:len(self._fold)]
            if not all(id(p) in self._fold for p in head):       # keep them a prefix of the buffers
                order = [p for p in self.buf.params if id(p) in self._fold]
                raise RuntimeError(f"Trainer: folded-gradient parameters are not a prefix of the flat buffers ({len(order)})")
        self._fold_end = self.buf.offsets[len(self._fold)] if self._fold else 0
        if _distributed(group):
            broadcast_(self.buf.flat_p, group)
        self.exp_avg = torch.zeros_like(self.buf.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.buf.flat_p)
        self._fold_m, self._fold_v = {}, {}
        for key, (p, n) in self._fold.items():
            shape = (n * (n + 1) // 2, p.shape[0])
            self._fold_m[key] = torch.zeros(shape, dtype=torch.float32, device=p.device)
            self._fold_v[key] = torch.zeros(shape, dtype=torch.float32, device=p.device)
        self._defer_fold = False    # inside step(): folded gradients are not unfolded into .grad
        self.steps = 0
        self.fused_loss = fused_loss
        self.direct_grads = direct_grads
        self._hook_of = {}          # id(param) -> its early all-reduce hook
        self._pending = []          # async all-reduces launched from hooks during backward
        self._folded = []           # (all-reduce, folded gradient [T,H], param, n_nodes) awaiting their unfold
        # flat ranges still to be reduced after backward; the unfolded gradient region of the
        # folded-Adam weights is never written or read inside step(), so it is not exchanged either
        self._rest = [(self._fold_end, self.buf.numel)]
        if _distributed(group) and overlap_floats:
            self._install_overlap_hooks(overlap_floats)

    def _install_overlap_hooks(self, min_floats):
        early = []
        for p, o in zip(self.buf.params, self.buf.offsets):
            if p.numel() >= min_floats:
                early.append((o, o + p.numel()))
                hook = self._make_hook(o, p.numel())
                p.register_post_accumulate_grad_hook(hook)
                self._hook_of[id(p)] = hook      # the direct-gradient path calls it itself
        rest, pos = [], self._fold_end
        for lo, hi in sorted(early):
            if hi <= pos:
                continue
            if lo > pos:
                rest.append((pos, lo))
            pos = hi
        if pos < self.buf.numel:
            rest.append((pos, self.buf.numel))
        self._rest = rest

    def _make_hook(self, offset, numel):
        def hook(param):
            flat = self.buf.flat_g[offset:offset + numel]
            if param.grad.data_ptr() != flat.data_ptr():      # autograd replaced the view: copy back
                flat.copy_(param.grad.reshape(-1))
                param.grad = flat.view(param.shape)
            if self._split_graph:                              # exchanged as one range after the backward graph
                return
            self._pending.append(all_reduce_sum_(flat, self.group, async_op=True))
        return hook

    def forward_backward(self, inputs, target, eps=None, _defer_fold=False):
        """inputs: positional arguments of model.forward; returns the detached loss (0-dim).
        loss_mode "mdn" trains a model whose forward returns (component means, gate logits, mu,
        logvar) and whose .loss() is the trait-gated mixture likelihood + KL.
        With `direct_grads` (default) the kernels that produce parameter gradients add them
        straight into the flat gradient buffer (ops.accumulate_into_grads) instead of going through
        one AccumulateGrad kernel per parameter."""
        if not self.direct_grads:
            return self._forward_backward(inputs, target, eps)
        exchange = bool(self._hook_of)
        self._defer_fold = bool(_defer_fold and self._fold)
        try:
            with ops.accumulate_into_grads(ready=self._grad_ready if (exchange and not self._split_graph) else None,
                                           folded=self._folded_grad if (exchange or self._fold) else None):
                return self._forward_backward(inputs, target, eps)
        finally:
            self._defer_fold = False

    def _grad_ready(self, param):
        hook = self._hook_of.get(id(param))
        if hook is not None:
            hook(param)

    def _folded_grad(self, param, gt, n_nodes):
        """Weight gradient of a symmetric-input first layer, still folded onto the upper triangle
        ([T, H], half the elements of the weight): that is what crosses the links.  The SUM commutes
        with writing each value to (i,j) and (j,i), which optimizer_step does once the transfer is in."""
        hooked, folded_adam = id(param) in self._hook_of, id(param) in self._fold
        if not hooked and not folded_adam:   # a small layer: unfold now, its range goes with the buckets
            call("mdnvae_tri_unfold", ptr(gt), gt.shape[1], n_nodes, gt.shape[1], 1, ptr(param.grad), 1, stream())
            return
        if not _distributed(self.group) or self._split_graph:
            work = None
            if not self._defer_fold and not self._split_graph:         # a caller outside step() may read .grad
                call("mdnvae_tri_unfold", ptr(gt), gt.shape[1], n_nodes, gt.shape[1], 1, ptr(param.grad), 1, stream())
        elif hooked:
            work = all_reduce_sum_(gt, self.group, async_op=True)
        else:                                # below the early-exchange size: summed where it is consumed
            work = "late"
        self._folded.append((work, gt, param, n_nodes))

    def _forward_backward(self, inputs, target, eps=None):
        with ops.arithmetic(wgrad=self.wgrad_arithmetic):
            return self._forward_backward_pass(inputs, target, eps)

    def _forward_backward_pass(self, inputs, target, eps=None):
        self._folded.clear()
        # inside step() the unfolded gradients of the folded-Adam weights are neither written nor read
        self.buf.zero_grad(self._fold_end if self._defer_fold else 0)
        if self.loss_mode == "mdn" and self.fused_loss:
            # rank-one component means: likelihood, KL and all their gradients from one pass over
            # the target, straight from the decoder's node vectors
            m = self.model
            nodes, gate_logits, mu, logvar = m(*inputs, eps=eps, nodes_only=True)
            # a target that is the very input buffer of a symmetric-input model has had its symmetry
            # checked by the encoder's packing pass: the likelihood can then stream it
            terms, g_nodes, g_logits, g_ls, g_mu, g_lv = ops.mdn_outer_loss_and_grads(
                nodes.detach(), target.reshape(target.shape[0], -1), gate_logits.detach(), m.log_sigma.detach(),
                mu.detach(), logvar.detach(), m.n_nodes,
                symmetric_x=bool(getattr(m, "symmetric_input", False)) and _same_buffer(target, inputs[0]))
            torch.autograd.backward([nodes, gate_logits, mu, logvar, m.log_sigma], [g_nodes, g_logits, g_mu, g_lv, g_ls])
            return terms[0]
        if self.fused_loss and self.loss_mode in ops.RECON_MODES and getattr(self.model, "rank_one_sum_decoder", False):
            # plain sum of outer products (the k-NN GCN decoder): loss, KL and gradients from the
            # node vectors, the reconstruction itself is never written
            m = self.model
            nodes, mu, logvar = m(*inputs, eps=eps, nodes_only=True)
            terms, g_nodes, g_mu, g_lv = ops.outer_sum_loss_and_grads(
                nodes, target.reshape(target.shape[0], -1), mu, logvar, self.loss_mode, m.n_nodes,
                symmetric_x=bool(getattr(m, "symmetric_input", False)) and _same_buffer(target, inputs[0]))
            torch.autograd.backward([nodes, mu, logvar], [g_nodes, g_mu, g_lv])
            return terms[0]
        out = self.model(*inputs, eps=eps)
        if self.loss_mode == "mdn":
            comp, gate_logits, mu, logvar = out
            loss = self.model.loss(comp, gate_logits, target, mu, logvar)
            loss.backward()
            return loss.detach()
        recon, mu, logvar = out[0], out[1], out[2]
        if self.fused_loss:
            terms, g_recon, g_mu, g_lv = ops.vae_loss_and_grads(recon.detach(), target, mu.detach(), logvar.detach(),
                                                                self.loss_mode)
            torch.autograd.backward([recon, mu, logvar], [g_recon, g_mu, g_lv])
            return terms[0]
        loss = ops.vae_loss(recon, target, mu, logvar, self.loss_mode)
        loss.backward()
        return loss.detach()

    def optimizer_step(self):
        for lo, hi in self._rest:
            allreduce_sum(self.buf.flat_g[lo:hi], self.group)
        for w in self._pending:
            w.wait()
        self._pending.clear()
        self.steps += 1
        self._apply_adam(None)

    def _apply_adam(self, dev_state):
        """Adam over the flat buffers, and over the folded-gradient weights in their own layout.
        dev_state = (step counter, coefficient pair) on the device for a captured step, else None."""
        folded = []
        for work, gt, param, n_nodes in self._folded:
            if work == "late":
                all_reduce_sum_(gt, self.group)
            elif work is not None:
                work.wait()
            if id(param) in self._fold:
                folded.append((gt, param, n_nodes))
            else:
                call("mdnvae_tri_unfold", ptr(gt), gt.shape[1], n_nodes, gt.shape[1], 1, ptr(param.grad), 1, stream())
        self._folded.clear()
        if len(folded) != len(self._fold):
            raise RuntimeError("Trainer: a symmetric-input weight received no folded gradient in this step (a backward "
                               "pass outside Trainer.forward_backward?); build the trainer with folded_adam=False")
        lo = self._fold_end
        args = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), 1.0, stream())
        rest = (ptr(self.buf.flat_p[lo:]), ptr(self.buf.flat_g[lo:]), ptr(self.exp_avg[lo:]), ptr(self.exp_avg_sq[lo:]),
                self.buf.numel - lo)
        if dev_state is None:
            call("mdnvae_adam_step", *rest, self.steps, *args)
        else:
            call("mdnvae_adam_step_dev", *rest, dev_state[0].data_ptr(), ptr(dev_state[1]), *args)
        for gt, param, n_nodes in folded:
            call("mdnvae_adam_step_folded", ptr(param.data), ptr(gt), ptr(self._fold_m[id(param)]),
                 ptr(self._fold_v[id(param)]), param.shape[0], n_nodes, max(self.steps, 1),
                 ptr(dev_state[1]) if dev_state is not None else None, *args)

    def step(self, inputs, target, eps=None):
        loss = self.forward_backward(inputs, target, eps, _defer_fold=True)
        self.optimizer_step()
        return loss

    # ---- one step as a HIP graph ---------------------------------------------------------
    def capture(self, inputs, target, eps):
        """Record zero_grad -> forward -> loss -> backward -> Adam for this batch shape once into a
        HIP graph; `replay` then runs a step as a single graph launch.  A step is a hundred or
        more kernel launches, and at the scripts' batch sizes (function/VAE_CNN_NN.py:53, batch
        256) the host cannot issue them as fast as the GPU finishes them; the replayed graph
        removes that limit.  Tensor inputs, the target and eps are copied into static buffers
        (the graph reads those addresses); other inputs (the adjacency list) are baked in.  The
        warm-up and capture passes leave parameters, Adam state and step count untouched.
        With data-parallel ranks the step is recorded as TWO graphs split at the exchange: zero_grad ->
        forward -> loss -> backward, then (outside any graph) the SUM all-reduce of the gradient buffer and
        of the folded first-layer gradients, then Adam.  Nothing is exchanged during backward in that form."""
        if eps is None:
            raise ValueError("Trainer.capture: eps must be given (the graph cannot draw fresh noise itself)")
        dev = self.buf.flat_p.device
        self._g_inputs = tuple(t.detach().clone() if isinstance(t, torch.Tensor) else t for t in inputs)
        # an autoencoding target (the input buffer itself) stays the input buffer in the graph
        self._g_target = self._g_inputs[0] if _same_buffer(target, inputs[0]) else target.detach().clone()
        self._g_eps = eps.detach().clone()
        self._g_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._g_coef = torch.zeros(2, dtype=torch.float32, device=dev)
        state = [self.buf.flat_p, self.exp_avg, self.exp_avg_sq] + list(self._fold_m.values()) + \
            list(self._fold_v.values())
        keep = [t.clone() for t in state]

        def backward_part():
            return self.forward_backward(self._g_inputs, self._g_target, self._g_eps, _defer_fold=True)

        def adam_part():
            self._apply_adam((self._g_step, self._g_coef))

        self._split_graph = split = _distributed(self.group)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):                      # first calls: library attributes, allocator pools
                    backward_part()
                    adam_part()
            torch.cuda.current_stream(dev).wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            self._graph_adam = None
            with torch.cuda.graph(self._graph):
                self._g_loss = backward_part()
                if not split:
                    adam_part()
            if split:
                # the folded gradients the first graph leaves (its own memory, alive while this list is) are
                # what the exchange sums and what the second graph reads
                self._g_folded = list(self._folded)
                self._graph_adam = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_adam, pool=self._graph.pool()):
                    adam_part()
        finally:
            self._split_graph = False
        for dst, src in zip(state, keep):
            dst.copy_(src)
        self._g_step.fill_(self.steps)
        self._g_step_holds = self.steps      # the count the device counter holds
        return self

    def replay(self, inputs, target, eps):
        """One captured step on new data of the captured shapes; returns the loss tensor of the
        graph (overwritten by the next replay)."""
        for dst, src in zip(self._g_inputs, inputs):
            if isinstance(dst, torch.Tensor):
                dst.copy_(src)
        if self._g_target is not self._g_inputs[0]:
            self._g_target.copy_(target)
        self._g_eps.copy_(eps)
        if self._g_step_holds != self.steps:     # eager steps since the last replay moved the host count on
            self._g_step.fill_(self.steps)
        self._graph.replay()
        if self._graph_adam is not None:
            allreduce_sum(self.buf.flat_g[self._fold_end:], self.group)
            for _, gt, _, _ in self._g_folded:
                all_reduce_sum_(gt, self.group)
            self._graph_adam.replay()
        self.steps += 1
        self._g_step_holds = self.steps
        return self._g_loss

    def hinge_step(self, fc, sc, eps_with=None, eps_without=None, generator=None, margin=0.01):
        """Conditional-VAE step of function_FC_SC/SC-FC.py:134-167: one forward with the FC input,
        one with the FC rows of a random half of the batch zeroed, and
        loss = loss_with + relu(loss_with - loss_without + margin).  Returns the detached loss."""
        self.buf.zero_grad()
        B = fc.shape[0]
        idx = torch.randperm(B, device=fc.device, generator=generator)[: B // 2]
        fc_without = fc.clone()
        fc_without[idx] = 0.0
        with ops.arithmetic(wgrad=self.wgrad_arithmetic):
            recon, mu, logvar = self.model(fc, sc, eps=eps_with)
            loss_with = ops.vae_loss_terms(recon, sc, mu, logvar, "mse")[0]
            recon_no, mu_no, logvar_no = self.model(fc_without, sc, eps=eps_without)
            loss_without = ops.vae_loss_terms(recon_no, sc, mu_no, logvar_no, "mse")[0]
            loss = loss_with + torch.relu(loss_with - loss_without + margin)
            loss.backward()
        self.optimizer_step()
        return loss.detach()

    def train_epoch(self, data, batch_size=256, extra=(), target=None, shuffle=True, generator=None,
                    eps_generator=None, graph=False):
        """One pass over `data` [n, D] (device resident) in batches of `batch_size`, last batch
        ragged, order reshuffled when `shuffle` (function/VAE_CNN_NN.py:53-55, 209-222).  The
        per-step losses are accumulated on the device and read back once; returns the epoch
        metric of the scripts, sum of losses / number of subjects (with data-parallel ranks: of
        this rank's shard).  `extra(idx)` may be a callable giving per-batch extra inputs (e.g.
        traits) or a tuple of fixed extra inputs (the adjacency list).  `graph`: full batches run
        as replays of one captured HIP graph (see `capture`; needs `eps_generator`); the ragged last batch is stepped call by call."""
        if graph and eps_generator is None:
            raise ValueError("train_epoch(graph=True) needs eps_generator: a captured step cannot draw its own noise")
        n = data.shape[0]
        order = torch.randperm(n, device=data.device, generator=generator) if shuffle else \
            torch.arange(n, device=data.device)
        latent = self.exp_avg.new_zeros(())                     # running loss on the device
        self.model.train()
        for lo in range(0, n, batch_size):
            idx = order[lo:lo + batch_size]
            x = data.index_select(0, idx)
            y = x if target is None else target.index_select(0, idx)
            more = extra(idx) if callable(extra) else tuple(extra)
            eps = None
            if eps_generator is not None:
                eps = torch.randn(x.shape[0], self._latent_dim(), device=x.device, generator=eps_generator)
            if graph and x.shape[0] == batch_size:
                if getattr(self, "_graph_batch", None) != batch_size:
                    self.capture((x,) + tuple(more), y, eps)
                    self._graph_batch = batch_size
                latent += self.replay((x,) + tuple(more), y, eps)
            else:
                latent += self.step((x,) + tuple(more), y, eps=eps)
        total = float(latent.item())
        check = getattr(self.model, "check_inputs", None)
        if check is not None:
            check()
        return total / n

    def _latent_dim(self):
        m = self.model
        head = getattr(m, "fc_mean", None) or getattr(m, "fc21", None) or getattr(m, "fc_mu")
        return head.out_features
