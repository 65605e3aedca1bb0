"""The SiD-LSG inner step (the hot path): one fake-score update + one generator update.

Restates training/sid_training_loop.py:383-571 for HipUNet2DCondition networks:
  phase A (:389-462)  x_hat = G(z) without grad -> eps_psi at guidance kappa1 on the CFG batch ->
                      fake-score MSE (closed-form grad) -> backward through psi -> fused nan_to_num+Adam
  phase B (:468-549)  x_hat = G(z) with grad -> y_fake (psi frozen, kappa2) and y_real (teacher phi, kappa4) on
                      ONE shared noisy CFG batch -> SiD-LSG loss (closed-form grads) -> backward through psi, phi
                      (data gradients only) and G -> fused nan_to_num+Adam+EMA
Differences from the reference that do NOT change the math (DESIGN.md "Step"):
  * inputs are explicit (z, noise, t, text states): RNG order is owned by the caller (training_loop);
  * per-sample NaN filtering happens inside the loss kernels (no host sync);
  * x_t, the timestep embedding input and the text states are built once and shared by psi and phi in phase B;
  * gradients accumulate into flat buffers; the data-parallel exchange is a few large RCCL all-reduces whose
    mean is folded into the optimizer kernel;
  * EMA, bf16 weight refresh and zero_grad are fused into the optimizer kernel.
"""
import os

import torch

from . import ops
from .sd_util import hip_denoise, hip_generate, hip_prepare_denoise


class _SegmentedUpdate:
    """Optimizer step of one network, issued segment by segment from inside its backward (HipUNet2DCondition.grad_segments:
    up blocks + head, mid block, the deep down blocks, the rest).  As soon as the backward has passed a segment, that segment's gradient exchange
    (world > 1) and its slice of the fused nan_to_num + Adam (+ EMA) + bf16-copy kernel run on the optimizer stream beside the
    MFMA-bound backward of the earlier layers: the HBM-bound optimizer (25 GB of traffic per network, ~5.5 ms on an MI355X)
    leaves the critical path except for the last segment.  Elementwise kernel, disjoint ranges: bit-identical to one launch
    over the whole buffer after the backward."""

    def __init__(self, step, net, opt):
        self.step, self.net, self.opt = step, net, opt
        self.segs = net.grad_segments()
        self.use_ema = False

    def arm(self, ema_beta):
        """Before the FORWARD of the last accumulation round (the markers are placed by the forward)."""
        if not self.opt.external_scalars:
            self.opt.begin_step(ema_beta)
        self.use_ema = ema_beta is not None and self.opt.ema is not None
        self.net.set_grad_ready_callback(self._ready)

    def _ready(self, k):
        st, net = self.step, self.net
        cur = torch.cuda.current_stream()
        if st.exchange:
            st.reducer.start_segment(net.flat_grads, self.segs[k])    # communication stream, after cur + weight-gradient streams
        st.opt_stream.wait_stream(cur)
        for s in ops.grad_streams(net.flat_grads.device):
            st.opt_stream.wait_stream(s)
        with torch.cuda.stream(st.opt_stream):
            if st.exchange:
                st.reducer.wait()                                     # orders the optimizer stream after the collectives
            for lo, hi in self.segs[k]:
                self.opt.launch_range(lo, hi, use_ema=self.use_ema)

    def start_last(self):
        """After the backward: the remaining segment (still on the optimizer stream: it runs beside whatever the caller
        enqueues next on the compute stream)."""
        self.net.set_grad_ready_callback(None)
        self._ready(len(self.segs) - 1)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.step.opt_stream)
        self.net.refresh_compute_weights(cast=False)                  # backward-data operands (transposed bf16 weights)


class SiDStep:
    def __init__(self, G, fake_score, true_score, G_ema, scheduler, opt_fake, opt_G, *, alpha=1.0, cfg_train_fake=1.0,
                 cfg_eval_fake=1.0, cfg_eval_real=1.0, loss_scaling=1.0, loss_scaling_G=1.0, batch_gpu_total=1,
                 init_timestep=625, reducer=None, world_size=1):
        self.G, self.psi, self.phi, self.G_ema = G, fake_score, true_score, G_ema
        self.sched, self.opt_fake, self.opt_G = scheduler, opt_fake, opt_G
        self.alpha, self.k1, self.k2, self.k4 = float(alpha), float(cfg_train_fake), float(cfg_eval_fake), float(cfg_eval_real)
        self.ls, self.lsg, self.bgt = float(loss_scaling), float(loss_scaling_G), int(batch_gpu_total)
        self.init_timestep = int(init_timestep)
        self.reducer, self.world = reducer, world_size
        # gradients are exchanged when there is more than one rank -- or when the reducer was built to run its collectives
        # on a single rank too (FlatGradReducer(min_world=1): how the RCCL path is exercised on a one-GPU box)
        self.exchange = reducer is not None and (world_size > 1 or getattr(reducer, 'min_world', 2) <= 1)
        self.overlap_g = os.environ.get('SIDLSG_OVERLAP_G', '1') != '0'    # A/B switch; results are identical either way
        # Phase B evaluates the teacher and the fake-score network on the SAME noisy CFG batch (identical layer shapes): the
        # teacher runs on a second HIP stream (forward, and through autograd its data-gradient backward), so the two
        # networks' kernels share the chip wherever one of them cannot fill 256 CUs (16x16 / 8x8 stages, split-K tails).
        self.side = None
        if os.environ.get('SIDLSG_TEACHER_STREAM', '1') != '0' and torch.cuda.is_available():     # A/B switch (+2 % images/s on MI355X)
            self.side = ops.side_stream(G.flat_params.device)
        # Grouped frozen passes (include/sidlsg_hip.h "grouped launches"): the two networks run as ONE pass over the stacked batch
        # [psi's CFG batch ; phi's CFG batch] -- every contraction / normalisation launch carries both parameter sets and picks
        # one per block, so the grids of the 16x16 / 8x8 stages, the time-embedding MLP and the 77-token K/V projections are
        # twice as large and the frozen passes issue half the launches (forward AND data-gradient backward).
        # Measured on one MI355X.  Round 4 (profiles/r04_grouped_ab.md): the grouped pass alone 1.10x faster than the two networks one
        # after the other; inside the step a clear win for small rounds (batch_gpu 1: 134 -> 113 ms, 2: 131 -> 113) and a 1-2 % loss at
        # batch_gpu 4 / 8 against the two-STREAM path.  Round 5, after the attention renumbering / rotated walk and the deferred
        # reductions (profiles/r05_grouped_ab.txt, five in-session alternations at batch_gpu 8): 209.4 -> 203.5 ms (-2.8 %), and with the
        # gradient exchange forced (RCCL world 1, `bench.py --force-exchange`) 221.8 -> 210.4 -- the joint pass halves the launches of
        # 4 F per image, and what it gives up (psi's optimizer step hidden under the teacher's forward) is 4-5 ms.  $SIDLSG_GROUPED_FROZEN:
        # 1 / auto (default) = grouped wherever both networks allow it (_can_group: same architecture, bf16, no e4m3 copies), 0 = never
        # (two-stream path).
        self.grouped_mode = os.environ.get('SIDLSG_GROUPED_FROZEN', 'auto').lower()
        self.grouped = self.grouped_mode == '1' and self._can_group()
        # opt-in: optimizer steps issued segment-wise from inside the backward, on their own stream (_SegmentedUpdate).  Same
        # results; measured NEUTRAL on one MI355X (221.5 / 220.9 vs 221.7 / 220.8 ms per iteration): the trace shows 3.2 of
        # the 4.8-5.5 ms of each optimizer kernel moving under the backward, and the kernels it then shares HBM with
        # (weight gradients, split-K reductions, GroupNorm backward) slowing down by as much.  Off by default.
        self.seg_opt = False
        self.opt_stream = None
        if os.environ.get('SIDLSG_SEG_OPT', '0') == '1' and torch.cuda.is_available():
            self.enable_segmented_optimizer()
        # phase B's generator forward issued before phase A, on the side stream (_early_generator_forward); A/B switch, same results
        self.early_gfwd = os.environ.get('SIDLSG_EARLY_GFWD', '1') != '0'
        self._graphs, self._graph_warm = {}, False
        opt_fake.grad_scale = opt_G.grad_scale = 1.0 / world_size     # DDP mean, folded into the optimizer kernel
        opt_fake.attach(ema=None, w16=fake_score.flat_w16, owner=fake_score)
        opt_G.attach(ema=(G_ema.flat_params if (G_ema is not None and G_ema is not G) else None), w16=G.flat_w16, owner=G)
        if not (G.compute_dtype == fake_score.compute_dtype == true_score.compute_dtype):
            raise ValueError('G, fake_score and true_score must share one compute dtype (they share the noisy CFG batch)')
        self.phi.requires_grad_(False)

    def _use_grouped(self, batch):
        if self.grouped_mode == '0':
            return False
        return self._can_group()

    def _can_group(self):
        from .unet import HipUNet2DCondition
        a, b = self.psi, self.phi
        return (torch.cuda.is_available() and type(a) is HipUNet2DCondition and type(b) is HipUNet2DCondition and a.cfg == b.cfg
                and a.compute_dtype == b.compute_dtype == torch.bfloat16 and 'fp8' not in a._flat and 'fp8' not in b._flat)

    def enable_segmented_optimizer(self, on=True):
        self.seg_opt = bool(on)
        self.opt_stream = torch.cuda.Stream(self.G.flat_params.device) if on else None

    def _init_t(self, n, device):
        return torch.full((n,), self.init_timestep, device=device, dtype=torch.long)

    # ---- phase A: fake-score network -----------------------------------------------------------
    def fake_round(self, r):
        """r: dict(z, noise, t, cond, uncond) (fp32 NCHW / int64 / bf16 text states)."""
        # (fp8_forward: the e4m3 forward copies of a network converted with enable_fp8_weights(frozen_passes_only=True) --
        # BASELINE.json configs[4]; a no-op otherwise)
        with torch.no_grad(), self.G.fp8_forward():                                 # :406-411
            images = hip_generate(self.G, r['z'], r['cond'], self._init_t(len(r['z']), r['z'].device), self.sched)
        prep = hip_prepare_denoise(images, r['noise'], r['t'], r['cond'], r.get('uncond'), self.sched, self.k1 != 1,
                                   act_dtype=self.psi.compute_dtype)
        eps = hip_denoise(self.psi, prep, self.k1, predict_x0=False)                # :418-421
        loss = ops.sid_fake_score_loss(eps, r['noise'], self.ls / self.bgt)         # :423-445
        loss.backward()                                                             # :449-450
        return loss.detach()

    def _early_generator_forward(self, r, overlap):
        """Phase B's generator forward x_hat = G(z) (with grad) reads nothing phase A writes -- G is only updated at the end of
        phase B -- so it is issued on the side stream BEFORE phase A and runs beside the fake-score network's forward and
        backward (its batch is half of theirs: alone it leaves the 16x16 / 8x8 stages even emptier than they do).  Autograd
        runs G's backward on the same stream at the end of phase B; the caller's stream is synchronised with it by the engine
        when loss.backward() returns (as for the teacher).  Returns (images, event).
        (Issuing the teacher's evaluation of x_hat here too -- it depends on nothing of phase A either -- measured neutral,
        216.9 vs 216.5 ms: it overlaps with the fake-score network's pass of phase B just as well.)"""
        self.G.requires_grad_(True)                                                 # :468
        if overlap:     # the markers of the segment-wise exchange are placed by the forward
            segs = self.G.grad_segments()
            self.G.set_grad_ready_callback(lambda k: self.reducer.start_segment(self.G.flat_grads, segs[k]))
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        for t in (r['z'], r['cond']):
            t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            images = hip_generate(self.G, r['z'], r['cond'], self._init_t(len(r['z']), r['z'].device), self.sched)
            ev = torch.cuda.Event()
            ev.record()
        return images, ev

    def fake_backward(self, rounds, seg=None, overlap_exchange=False, keep_G=False):
        """Forward/backward of phase A over all accumulation rounds; leaves the gradients in psi.flat_grads (seg: the
        segments the backward of the last round has passed are already being exchanged / updated)."""
        if not keep_G:       # (keep_G: phase B's forward of G is already in flight with grad; phase A's own G pass is under no_grad)
            self.G.requires_grad_(False)
        self.psi.requires_grad_(True)                                               # :389
        loss = None
        overlap = seg is None and overlap_exchange and self.exchange and self.overlap_g
        segs = self.psi.grad_segments() if overlap else None
        for i, r in enumerate(rounds):
            if seg is not None and i == len(rounds) - 1:
                seg.arm(None)
            if overlap and i == len(rounds) - 1:
                # last accumulation round: a segment of psi's flat gradient is exchanged as soon as the backward has passed it
                # (the same scheme as for the generator in generator_update)
                self.psi.set_grad_ready_callback(lambda k: self.reducer.start_segment(self.psi.flat_grads, segs[k]))
            loss = self.fake_round(r)
        if overlap:
            self.psi.set_grad_ready_callback(None)
            self.reducer.start_segment(self.psi.flat_grads, segs[-1])
        self.psi.requires_grad_(False)                                              # :455
        return loss

    def fake_update(self, rounds):
        if self.seg_opt:
            seg = _SegmentedUpdate(self, self.psi, self.opt_fake)
            loss = self.fake_backward(rounds, seg)
            seg.start_last()
            seg.join()
            return loss
        loss = self.fake_backward(rounds)
        self._optimizer_step(self.psi, self.opt_fake, ema_beta=None)                # :458-462
        return loss

    # ---- phase B: generator --------------------------------------------------------------------
    def generator_round(self, r, before_fake_eval=None, pre=None):
        if pre is not None:                                                         # issued before phase A (_early_generator_forward)
            images, ev = pre
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            images.record_stream(cur)
        else:
            images = hip_generate(self.G, r['z'], r['cond'], self._init_t(len(r['z']), r['z'].device), self.sched)  # :488-491
        guided = (self.k2 != 1) or (self.k4 != 1)
        prep = hip_prepare_denoise(images, r['noise'], r['t'], r['cond'], r.get('uncond'), self.sched, guided,
                                   act_dtype=self.psi.compute_dtype)
        k2 = self.k2 if guided else 1.0
        k4 = self.k4 if guided else 1.0
        if self._use_grouped(len(r['z'])):
            # one grouped pass over [psi's batch ; phi's batch] (:494-506: two sid_sd_denoise calls on identical inputs)
            if before_fake_eval is not None:
                before_fake_eval()
            eps_f, eps_r = self.psi.forward_pair(self.phi, prep.xin, prep.tt, prep.ctx)
            cd = self.psi.compute_dtype
            y_fake = ops.cfg_x0(eps_f, prep.xt, prep.s0, prep.s1, k2, True, cd)     # u + k (c - u), then x0 (sid_sd_util.py:264-272)
            y_real = ops.cfg_x0(eps_r, prep.xt, prep.s0, prep.s1, k4, True, cd)
            loss = ops.sid_generator_loss(images, y_real, y_fake, self.alpha, self.lsg / self.bgt)   # :508-530
            loss.backward()                                                             # :532-533
            return loss.detach()
        # teacher first: neither G's forward nor phi's reads psi, so a pending psi gradient exchange / optimizer step
        # (before_fake_eval) overlaps with them.  y_real and y_fake are independent: the order does not change the math.
        if self.side is None:
            y_real = hip_denoise(self.phi, prep, k4, predict_x0=True)               # :503-506
        else:
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            for t in (prep.xin, prep.xt, prep.s0, prep.s1, prep.ctx, prep.tt):
                t.record_stream(self.side)
            with torch.cuda.stream(self.side):
                y_real = hip_denoise(self.phi, prep, k4, predict_x0=True)
            y_real.record_stream(cur)
        if before_fake_eval is not None:
            before_fake_eval()
        with self.psi.fp8_forward():        # psi is frozen here: data gradient only (through its bf16 backward-data operands)
            y_fake = hip_denoise(self.psi, prep, k2, predict_x0=True)               # :496-499
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        loss = ops.sid_generator_loss(images, y_real, y_fake, self.alpha, self.lsg / self.bgt)   # :508-530
        loss.backward()                                                             # :532-533
        return loss.detach()

    def generator_update(self, rounds, ema_beta=None, before_fake_eval=None, pre=None):
        self.G.requires_grad_(True)                                                 # :468
        self.psi.requires_grad_(False)
        loss = None
        if self.seg_opt:
            seg = _SegmentedUpdate(self, self.G, self.opt_G)
            for i, r in enumerate(rounds):
                if i == len(rounds) - 1:
                    seg.arm(ema_beta)
                loss = self.generator_round(r, before_fake_eval if i == 0 else None)
            self.G.requires_grad_(False)                                            # :538
            seg.start_last()                                                        # :541-565
            seg.join()
            return loss
        overlap = self.exchange and self.overlap_g
        segs = self.G.grad_segments() if overlap else None
        for i, r in enumerate(rounds):
            if overlap and i == len(rounds) - 1 and pre is None:
                # last accumulation round (what DDP does outside no_sync): a segment of the flat gradient is exchanged as
                # soon as the backward has passed it, while the earlier layers' backward is still running
                self.G.set_grad_ready_callback(lambda k: self.reducer.start_segment(self.G.flat_grads, segs[k]))
            loss = self.generator_round(r, before_fake_eval if i == 0 else None, pre if i == 0 else None)
        self.G.requires_grad_(False)                                                # :538
        if overlap:
            self.G.set_grad_ready_callback(None)
            self.reducer.start_segment(self.G.flat_grads, segs[-1])
        self._optimizer_step(self.G, self.opt_G, ema_beta=ema_beta, started=overlap)   # :541-565
        return loss

    # ---- optimizer + data-parallel exchange ------------------------------------------------------
    def _optimizer_step(self, net, opt, ema_beta, started=False):
        if self.exchange:
            if not started:
                self.reducer.start(net.flat_grads)  # few large all-reduce(SUM) on the comm stream
            self.reducer.wait(tag='fake_score' if net is self.psi else 'G')
        opt.step(ema_beta=ema_beta)                 # nan_to_num, /world, Adam, EMA, bf16 copy, zero_grad: one kernel
        net.refresh_compute_weights(cast=False)     # backward-data operands (transposed bf16 weights)

    def iteration(self, inputs, ema_beta=None):
        """inputs: dict(A=[rounds], B=[rounds]).  Returns (loss_fake, loss_G) as device scalars.

        Same result as fake_update(); generator_update(), but the psi gradient all-reduce + optimizer step are issued
        after phase A's backward and only WAITED for right before psi is evaluated in phase B, i.e. they overlap with
        the generator forward and the teacher forward of the first phase-B round."""
        if self.seg_opt:
            # psi: segments 0 / 1 are exchanged + updated during its backward, the last one right after it on the optimizer
            # stream -- beside the generator forward and the teacher forward; joined right before psi is evaluated
            seg = _SegmentedUpdate(self, self.psi, self.opt_fake)
            lf = self.fake_backward(inputs['A'], seg)
            seg.start_last()
            lg = self.generator_update(inputs['B'], ema_beta=ema_beta, before_fake_eval=seg.join)
            return lf, lg
        pre = None
        if self.early_gfwd and self.side is not None and len(inputs['B']) == 1:
            pre = self._early_generator_forward(inputs['B'][0], self.exchange and self.overlap_g)
        lf = self.fake_backward(inputs['A'], overlap_exchange=True, keep_G=pre is not None)
        overlap = self.exchange
        if overlap and not self.overlap_g:
            self.reducer.start(self.psi.flat_grads)

        def finish_fake():
            self._optimizer_step(self.psi, self.opt_fake, ema_beta=None, started=overlap)
        lg = self.generator_update(inputs['B'], ema_beta=ema_beta, before_fake_eval=finish_fake, pre=pre)
        return lf, lg

    # ---- the same iteration as ONE HIP graph ---------------------------------------------------------------------------
    @staticmethod
    def _signature(inputs, ema_beta):
        return (tuple((ph, tuple(tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(r.items()) if v is not None) for r in inputs[ph]))
                      for ph in ('A', 'B')), ema_beta is None)

    def iteration_graphed(self, inputs, ema_beta=None):
        """iteration() replayed from a captured HIP graph: the ~6500 launches of one iteration (all three streams, both
        optimizer kernels, for world > 1 the RCCL all-reduces on the communication stream) become one hipGraphLaunch, so the
        host cost of an iteration is the input copies + two scalar uploads instead of ~140 ms of Python enqueueing.
        The C ABI is allocation- and sync-free (include/sidlsg_hip.h), torch's allocations inside the capture come from the
        graph's private pool, and everything that varies between iterations lives in device memory: the inputs (copied
        into static buffers) and the optimizers' scalars (FusedAdamEMA.begin_step).
        Call protocol: the first call runs eagerly (lazy initialisations: workspaces, hipBLASLt handles of torch ops); the
        second call with a given input signature captures and replays; later calls replay.  Returns (loss_fake, loss_G) as
        STATIC device scalars that the next replay overwrites -- read them before the next call.
        Requires the fused optimizers (mode 1); a foreign optimizer / DDP wrapper (mode 2) has host logic per step."""
        if not self._graph_warm:
            self._graph_warm = True
            return self.iteration(inputs, ema_beta=ema_beta)
        sig = self._signature(inputs, ema_beta)
        g = self._graphs.get(sig)
        if g is None:
            static = {ph: [{k: (v.clone() if v is not None else None) for k, v in r.items()} for r in inputs[ph]] for ph in ('A', 'B')}
            self.opt_fake.begin_step(None)
            self.opt_G.begin_step(ema_beta)
            self.opt_fake.external_scalars = self.opt_G.external_scalars = True
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread polls events (hipEventQuery) at any time: under the default
            # 'global' capture mode that call, made by ANOTHER thread, aborts the process ("operation not permitted when
            # stream is capturing"); 'thread_local' confines the check to this thread.  Single-process runs keep 'global'.
            dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
            if dist_on:
                # Let the process group's watchdog retire the collectives of the eager iterations before the capture starts.
                # They are complete (synchronize above), but the watchdog only drops them on its next pass (every ~100 ms), and
                # until then it polls their end events -- which this HIP runtime refuses with hipErrorCapturedEvent as soon as
                # the stream they were recorded on (the backend's own) has joined a capture, taking the process down from the
                # watchdog thread ("operation not permitted on an event last recorded in a capturing stream"; 1-2 in 10 runs
                # of tests/test_gpu_dist.py::test_rccl_collectives_inside_the_captured_iteration without this pause).
                # Why a pause and not a handshake: the works ARE complete, what has to happen is one pass of the watchdog's clean-up loop
                # (ProcessGroupNCCL::watchdogHandler sleeps kWatchdogThreadSleepMillis = 100 ms between passes and drops every completed
                # work in one pass), and this torch build exposes nothing to wait for that pass: the flight recorder
                # (_dump_nccl_trace) keeps no entries unless TORCH_NCCL_TRACE_BUFFER_SIZE was set before the group was created
                # (tools/nccl_trace_probe.py: 0 entries on the GPU box).  0.5 s = five watchdog periods, once per captured signature.
                import time
                time.sleep(0.5)
            try:
                with torch.cuda.graph(graph, capture_error_mode='thread_local' if dist_on else 'global'):
                    lf, lg = self.iteration(static, ema_beta=ema_beta)
            finally:
                self.opt_fake.external_scalars = self.opt_G.external_scalars = False
            g = self._graphs[sig] = dict(graph=graph, static=static, out=(lf, lg))
        else:
            for ph in ('A', 'B'):
                for dst, src in zip(g['static'][ph], inputs[ph]):
                    for k, v in src.items():
                        if v is not None:
                            dst[k].copy_(v, non_blocking=True)
            self.opt_fake.begin_step(None)
            self.opt_G.begin_step(ema_beta)
        g['graph'].replay()
        return g['out']
