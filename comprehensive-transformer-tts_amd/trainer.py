"""The train step of the reference's `train.py:102-125` as the product runs it on an MI355X:

    forward -> CompTransTTSLoss -> backward [-> gradient all-reduce] -> clip_grad_norm_(1.0) -> Adam (Noam lr) -> zero_grad

Host orchestration only; every kernel is behind `ops` / `kernels`.  What this module adds over a literal loop:

  * gradients live in `dp.FlatGradArena` (kernels accumulate straight into it), the optimizer is `dp.FlatAdam` (fused clip + Adam);
  * the step is replayed from hipGraphs (the model has no host syncs in training);
  * data parallel (world > 1): the backward pass is STAGED (`ops.CutRecorder`, cuts from `dp.stage_plan`), one hipGraph per stage,
    and after each stage the finished bucket of the arena is all-reduced on a side stream by `dp.BucketedReducer` while the next
    stage replays - DDP's bucketed overlap (train.py:58) without autograd hooks, so it survives graph capture;
  * `feed(device_buffer)`: the inputs can be views of one packed device buffer (`data.PackedBatch`), refreshed by a single
    device-to-device copy per step, so a prefetching loader (`data.Prefetcher`) can sit inside the timed loop.
"""
import torch

from . import ops
from .dp import FlatGradArena, FlatAdam, BucketedReducer, stage_plan


class TrainStep:
    def __init__(self, model, loss_fn, optim, model_args, world=1, group=None, use_graph=True, overlap=True, n_cuts=3,
                 step_no=50001, fused_optimizer=True, max_norm=1.0, adam_step=0, force_staged=False, graph_collectives=None,
                 always_reduce=False, grad_acc_step=1):
        """`model_args`: positional arguments of CompTransTTS.forward (static device tensors; the graphs read them in place).
        `optim`: loss.ScheduledOptim(..., capturable=True) - owns the Noam schedule and the device-resident lr.
        `adam_step`: Adam's bias-correction step count to start from (a resumed run: the restore step; `FlatAdam.load_state_dict`
        sets it from a checkpoint).  `force_staged`: stage the backward pass even with world == 1 (tests).
        `graph_collectives` (None = env CTTS_GRAPH_COLLECTIVES, default off): capture the WHOLE step - the backward stages, the bucketed
        all-reduces on their side stream (RCCL collectives are capturable) and the optimizer - as ONE hipGraph, so a replayed step has
        no host work at all between its stages; off: one graph per stage with eager collectives between the replays (works with every
        backend; the only mode gloo supports).  `always_reduce`: issue the collectives with world == 1 too (1-GPU RCCL tests)."""
        self.model, self.loss_fn, self.optim, self.world = model, loss_fn, optim, int(world)
        self.args = list(model_args)
        self.loss_inputs = [None, None] + list(self.args)
        self.step_no = step_no
        ops.set_grad_accumulation_fusion(True)          # kernels accumulate straight into the flat gradient arena
        self.arena = FlatGradArena(model.named_parameters())
        self.params = self.arena.params
        self.flat_grad = self.arena.flat
        self.use_graph = bool(use_graph)
        self.max_norm = max_norm
        self.grad_acc_step = max(1, int(grad_acc_step))
        self.g_opt_noclip = None
        import os as _os          # CTTS_BATCH_REPACK=0: every conv layer repacks its own data-gradient weight inside its backward (A/B switch)
        self._conv_weights = ([p for p in model.parameters() if p.dim() == 3 and p.requires_grad and ops._gemm_major(p) is not None]
                              if _os.environ.get("CTTS_BATCH_REPACK", "1") != "0" else [])
        self.fadam = None
        if fused_optimizer:
            oc = optim._optimizer.defaults
            self.fadam = FlatAdam(self.arena, optim.lr_tensor, betas=tuple(oc["betas"]), eps=oc["eps"],
                                  weight_decay=oc["weight_decay"], max_norm=max_norm, current_step=adam_step)
        self.staged = ((self.world > 1 or always_reduce) and overlap) or force_staged
        self.cut_names, stage_of = stage_plan(model, n_cuts)
        self.n_stages = len(self.cut_names) + 1
        if not self.staged:
            self.cut_names, self.n_stages = [], 1
            stage_of = lambda name: 0                                         # noqa: E731
        self.reducer = BucketedReducer(self.arena, stage_of, self.n_stages, world=self.world, group=group, always_reduce=always_reduce)
        if graph_collectives is None:
            import os
            graph_collectives = os.environ.get("CTTS_GRAPH_COLLECTIVES", "0") == "1"
        self.graph_collectives = bool(graph_collectives) and self.reducer.active
        self.graphs = None              # [stage graphs], then the optimizer graph
        self.g_opt = None
        self.g_all = None               # graph_collectives: the whole step in one graph
        self.loss_val = None
        self.static_buffer = None
        # the stream-K hand-off reports a failed wait into its workspace; watch that word (async copy every `probe_every` steps, checked
        # one step later: no sync on the hot path) and at the end of capture - a failed launch must not go unnoticed
        from . import kernels as _K
        self.ws_probe = _K.WorkspaceErrorProbe()
        self.probe_every = 100
        self._calls = 0

    # ---- pieces
    def _forward_loss(self):
        args = list(self.args)
        if isinstance(args[7], dict):
            args[7] = dict(args[7])                     # the model mutates p_targets like the reference does
        with ops.side_loss_scope():                       # the CTC forward-sum term may run beside the decoder (ops.mark_ready)
            out = self.model(*args, step=self.step_no)
            inputs = list(self.loss_inputs)
            inputs[9:11] = out[-2:]
            losses = self.loss_fn(inputs, out[:-2], self.step_no)
        self.loss_val = losses[0].detach()
        self.loss_terms = losses                        # the reference's 9-tuple (train.py:127-137 logs it), graph-resident under replay
        return losses[0]

    def _stages(self):
        """generator over backward stages; forward + loss + stage 0 run on the first next().  During every stage a kernels.PartialSink
        collects the deferred ordered sums into the arena (split-K weight gradients, bias / LayerNorm column sums) and finishes them
        with one launch per 24 sums at the END of the stage - before the stage's bucket is all-reduced / the optimizer reads it."""
        from . import kernels as _K
        try:
            if self._conv_weights:
                # one launch: the transposed / flipped taps every conv's data gradient needs; one more: the bf16 planes of the weights
                # the plane kernel consumes.  Valid for THIS step only - the finally below drops them also when the step raises
                # (a stale entry would silently give a later backward last step's weights: ADVICE r04)
                ops.prepare_dgrad_weights(self._conv_weights)
            rec = ops.CutRecorder(self.cut_names)
            with rec:
                loss = self._forward_loss()
            self.flat_grad.zero_()
            if self.grad_acc_step > 1:
                loss = loss * (1.0 / self.grad_acc_step)          # train.py:112
            sink = _K.PartialSink()
            gen = rec.backward_stages(loss)
            while True:
                prev = _K.set_partial_sink(sink)
                try:
                    s = next(gen)
                    ops.join_side()                      # autograd ran the side branches' backward on their streams: their partials are final
                    sink.flush()
                except StopIteration:
                    return
                finally:
                    _K.set_partial_sink(prev)
                yield s
        finally:
            ops.clear_dgrad_weights()

    def _clip_now(self):
        return self.step_no % self.grad_acc_step == 0        # train.py:118 (always true for grad_acc_step = 1)

    def _optimizer_step(self, clip=True):
        if self.fadam is not None:
            self.fadam.max_norm = self.max_norm if clip else 0.0          # 0 = no clipping (csrc/optim.hip)
            self.fadam.step()
            return
        self.arena.check_bound()
        if clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
        self.optim._optimizer.step()

    def _eager(self):
        for s in self._stages():
            self.reducer.launch(s)
        self.reducer.finish()
        self._optimizer_step(self._clip_now())

    # ---- checkpointing (train.py:190-200 saves {"model": ..., "optimizer": optimizer._optimizer.state_dict()})
    def optimizer_state_dict(self):
        """the optimizer half of a checkpoint in torch.optim.Adam's format over ALL model.parameters() - from the fused FlatAdam when
        it does the updates (then `optim._optimizer` never steps and its own state stays empty), else from torch's Adam"""
        return self.fadam.state_dict() if self.fadam is not None else self.optim._optimizer.state_dict()

    def load_optimizer_state_dict(self, sd):
        if self.fadam is not None:
            self.fadam.load_state_dict(sd)
        else:
            self.optim._optimizer.load_state_dict(sd)

    # ---- graph capture
    def capture(self, warmup=2):
        """Side effects: the `warmup` eager steps are REAL optimizer steps (parameters, moments and the Noam step counter advance by
        `warmup`, exactly as if the loop had run them) - capture right before the training loop, not in the middle of an evaluation."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.optim.update_learning_rate()
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.graph_collectives:
            # ONE graph: stages, collectives (forked onto the reducer's side stream inside the capture, joined by finish()) and optimizer
            if self.fadam is None or self.grad_acc_step > 1:
                raise RuntimeError("graph_collectives needs the fused optimizer (torch's foreach clip + Adam are not capture-safe here) "
                                   "and grad_acc_step = 1 (the whole-step graph has one optimizer node)")
            self.g_all = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_all, stream=side, capture_error_mode="thread_local"):
                self._eager()
            self.ws_probe.poll_and_check()
            return
        # every stage is captured on the SAME stream (autograd runs a node's backward on the stream of its forward) and into one pool;
        # thread_local capture mode: the RCCL watchdog / a prefetch thread may touch the runtime while this thread captures
        graphs, gen, pool = [], self._stages(), None
        for s in range(self.n_stages):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=side, capture_error_mode="thread_local"):
                got = next(gen)
            assert got == s
            pool = g.pool()
            graphs.append(g)
        for _ in gen:                                     # exhaust (clears the recorder)
            raise AssertionError("more backward stages than planned")
        self.graphs = graphs
        if self.fadam is not None:                        # torch's foreach clip + Adam are not capture-safe on strided parameters: eager
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=pool, stream=side, capture_error_mode="thread_local"):
                self._optimizer_step(True)
            if self.grad_acc_step > 1:                    # the steps between two clipping steps: same update without the clip coefficient
                self.g_opt_noclip = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt_noclip, pool=pool, stream=side, capture_error_mode="thread_local"):
                    self._optimizer_step(False)
        self.ws_probe.poll_and_check()                    # the warm-up steps ran every kernel of the step once

    # ---- inputs as views of one packed device buffer (data.PackedBatch layout)
    def bind_static_buffer(self, buf):
        self.static_buffer = buf

    def feed(self, device_buffer):
        """refresh every model input with ONE device-to-device copy (same PackedBatch layout as the static buffer)"""
        self.static_buffer.copy_(device_buffer, non_blocking=True)

    def check_kernels(self):
        """synchronous check of the workspace error words (end of an epoch / before a checkpoint); raises CttsError"""
        self.ws_probe.poll_and_check()

    def __call__(self):
        self._calls += 1
        if self._calls % self.probe_every == 1:
            self.ws_probe.check()                       # the copy started `probe_every` steps ago has long landed: no stall
        self.optim.update_learning_rate()               # host scalar -> device lr tensor (outside the graphs)
        if self.g_all is not None:
            self.g_all.replay()
        elif self.graphs is not None:
            for s, g in enumerate(self.graphs):
                g.replay()
                self.reducer.launch(s)
            self.reducer.finish()
            clip = self._clip_now()
            if self.g_opt is not None:
                (self.g_opt if (clip or self.g_opt_noclip is None) else self.g_opt_noclip).replay()
            else:
                self._optimizer_step(clip)
        else:
            self._eager()
        if self._calls % self.probe_every == 0:
            self.ws_probe.poll()
        self.step_no += 1
