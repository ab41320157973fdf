"""Block-wise OmniQuant calibration of one quantized decoder block, with optional data parallelism.

Restates the per-layer part of the reference's ``omniquant()`` (quantize/omniquant.py:366-605):
reorder statistics -> ``Reorder`` -> full-precision targets -> LET initialisation -> AdamW loop over
(epochs x samples) with temporary smoothing + weight fake-quant, MSE against the targets, loss
scaling, per-step ``free()`` of the activation statistics -> in-place smoothing/quantization ->
sequential propagation of the quantized activations (this is where the static activation scales
accumulate) -> ``register_scales_and_zeros`` -> eval mode.

Data parallelism (one process per GPU): the calibration samples are sharded over the ranks; every
optimizer step each rank runs one local batch and the LWC/LET gradients are averaged with one packed
all-reduce, so the parameter trajectories stay identical on all ranks.  The final propagation pass is
order dependent (running min/max or EMA over the samples, each sample quantized with the statistics
of its prefix) and is therefore replicated: the owner broadcasts each sample, every rank runs it, the
owner keeps the result.
"""
from __future__ import annotations

import math
import time
import os
from contextlib import nullcontext
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import functional as MF
from . import reorder_utils as R
from .dist import DPContext, all_reduce, allreduce_mean_grads, broadcast_from, sample_owner
from .int_linear import QuantLinear
from .int_matmul import QuantMatMul
from .omni_norm import OmniLayerNorm


@dataclass
class CalibConfig:
    epochs: int = 20
    nsamples: int = 128
    batch_size: int = 1
    let: bool = True
    lwc: bool = True
    let_lr: float = 5e-3
    lwc_lr: float = 1e-2
    wd: float = 0.0
    alpha: float = 0.5
    deactive_amp: bool = True
    reorder: str = ""                      # e.g. "012345": "0" enables clustering, "1".."5" the R stages
    n_clusters: Dict[str, int] = field(default_factory=lambda: {"R1": 32, "R2": 4, "R3": 4, "R4": 32, "R5": 32})
    weight_mix_quant: bool = False
    loss_size_coef: float = 0.0
    use_shift: bool = True
    aug_loss: bool = False                 # add the MSE against the fp block applied to the QUANTIZED model's inputs
    fused_adamw: bool = False              # torch's fused AdamW: no host synchronisation in GradScaler.step, but its
                                           # arithmetic differs in the last bits from the default implementation
    graph_step: bool = False               # replay forward + loss + backward of a step from one HIP graph (captured after
                                           # the first, eager step; the optimizer stays eager).  Same kernels on the same
                                           # data: the trajectory is that of the eager loop.  Single process, fixed batch
                                           # shape, not with weight_mix_quant (its phases change the step)
    graph_check_every: int = 64            # with graph_step: every so many steps one step is also run eagerly and compared
    autocast_passes: bool = True           # run the full-precision target pass and the final propagation pass under fp16
                                           # autocast whatever deactive_amp says, as the reference does
                                           # (quantize/omniquant.py:402, :583); False keeps them in the training precision


class LossScaler:
    """GradScaler wrapper returning the gradient norm, optionally averaging gradients over ranks
    between backward and unscale (reference: utils.NativeScalerWithGradNormCount, utils.py:26-52)."""

    def __init__(self, device_type: str = "cuda"):
        self._scaler = torch.amp.GradScaler(device_type, enabled=(device_type == "cuda" and torch.cuda.is_available()))

    def backward(self, loss):
        self._scaler.scale(loss).backward()

    def step(self, optimizer, parameters, dp: Optional[DPContext] = None, buf=None):
        parameters = list(parameters)
        if dp is not None and dp.active:
            buf = allreduce_mean_grads(parameters, dp, buf)
        self._scaler.unscale_(optimizer)
        norm = grad_norm(parameters)
        self._scaler.step(optimizer)
        self._scaler.update()
        return norm, buf

    def __call__(self, loss, optimizer, parameters, dp: Optional[DPContext] = None, buf=None):
        self.backward(loss)
        return self.step(optimizer, parameters, dp, buf)


def grad_norm(parameters, norm_type: float = 2.0):
    """Norm of the per-parameter gradient norms (reference utils.ampscaler_get_grad_norm, utils.py:11-24);
    the per-parameter norms come from one multi-tensor launch instead of one launch per parameter."""
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    dev = grads[0].device
    if all(g.device == dev for g in grads):
        per = torch.stack(torch._foreach_norm(grads, norm_type))
    else:
        per = torch.stack([torch.norm(g, norm_type).to(dev) for g in grads])
    return torch.norm(per, norm_type)


# ------------------------------------------------------------------------------------------------
# reorder: statistics hooks and the R1..R5 wiring of a block
# ------------------------------------------------------------------------------------------------
def register_hooks(layer, enable_R1, enable_R2, enable_R3, enable_R4, enable_R5):
    """Statistics hooks by module name (reference omniquant.py:167-210).  Works on a plain fp block
    (nn.LayerNorm / nn.Linear) and on a quantized block run with quantization off."""
    norms = (nn.LayerNorm, OmniLayerNorm)
    linears = (nn.Linear, QuantLinear)
    handlers = []
    for name, module in layer.named_modules():
        hook = None
        if enable_R1 and isinstance(module, norms) and "attn_layer_norm" in name:
            hook = R.layer_omax_hook
        if enable_R2 and isinstance(module, linears) and ("q_proj" in name or "k_proj" in name):
            hook = R.layer_omax_hook
        if enable_R3 and isinstance(module, linears) and "out_proj" in name:
            hook = R.layer_i0max_hook
        if enable_R4 and isinstance(module, norms) and "final_layer_norm" in name:
            hook = R.layer_omax_hook
        if enable_R5 and isinstance(module, linears) and "fc2" in name:
            hook = R.layer_i0max_hook
        if hook is not None:
            module.name = name
            handlers.append(module.register_forward_hook(hook))
    return handlers


def sync_reorder_stats(dp: Optional[DPContext]):
    """Running max/min gathered from a sample shard -> global (one MAX all-reduce per dictionary)."""
    if dp is None or not dp.active:
        return
    import torch.distributed as dist

    for store in (R.oc_maxmin_dict, R.ic_maxmin_dict):
        names = sorted(store)
        if not names:
            continue
        packed = torch.cat([torch.cat((store[n][0].float(), -store[n][1].float())) for n in names])
        all_reduce(packed, dist.ReduceOp.MAX, dp.group)
        off = 0
        for n in names:
            c = store[n][0].numel()
            store[n] = (packed[off:off + c].to(store[n][0].dtype), (-packed[off + c:off + 2 * c]).to(store[n][1].dtype))
            off += 2 * c


def reorder_block(qlayer, hidden_size, ffn_dim, num_heads, enable, n_clusters, dev):
    """R1..R5 (reference omniquant.py:31-165).  A disabled stage still wires an identity index and
    equal-size clusters, and the identity gather still runs in the layer norms."""
    a = qlayer.self_attn
    indices = {}

    def even(total, k):
        return (np.ones(int(k), dtype=int) * (total / k)).astype(int)

    if not enable["R1"]:
        idx, counts = torch.arange(hidden_size).to(dev), even(hidden_size, n_clusters["R1"])
    else:
        fmax, fmin = R.oc_maxmin_dict["self_attn_layer_norm"]
        idx, counts = R.tensor_calc_reorder_index(fmax, fmin, n_clusters["R1"])
    R.R1_reorder(qlayer.self_attn_layer_norm, a.q_proj, a.k_proj, a.v_proj, idx, counts)
    indices["R1"] = idx

    if not enable["R2"]:
        idx, counts = torch.arange(hidden_size).to(dev), even(hidden_size, n_clusters["R2"] * num_heads)
    else:
        qmax, qmin = R.oc_maxmin_dict["self_attn.q_proj"]
        kmax, kmin = R.oc_maxmin_dict["self_attn.k_proj"]
        idx, counts = R.tensor_calc_reorder_index([qmax, kmax], [qmin, kmin], n_clusters["R2"], num_heads)
    R.R2_reorder(a.q_proj, a.k_proj, a.qkt_matmul, idx, counts)
    indices["R2"] = idx

    if not enable["R3"]:
        idx, counts = torch.arange(hidden_size).to(dev), even(hidden_size, n_clusters["R3"] * num_heads)
    else:
        fmax, fmin = R.ic_maxmin_dict["self_attn.out_proj"]
        idx, counts = R.tensor_calc_reorder_index(fmax, fmin, n_clusters["R3"], num_heads)
    R.R3_reorder(a.v_proj, a.pv_matmul, a.out_proj, idx, counts)
    indices["R3"] = idx

    if not enable["R4"]:
        idx, counts = torch.arange(hidden_size).to(dev), even(hidden_size, n_clusters["R4"])
    else:
        fmax, fmin = R.oc_maxmin_dict["final_layer_norm"]
        idx, counts = R.tensor_calc_reorder_index(fmax, fmin, n_clusters["R4"])
    R.R4_reorder(qlayer.final_layer_norm, qlayer.fc1, idx, counts)
    indices["R4"] = idx

    if not enable["R5"]:
        idx, counts = torch.arange(ffn_dim).to(dev), even(ffn_dim, n_clusters["R5"])
    else:
        fmax, fmin = R.ic_maxmin_dict["fc2"]
        idx, counts = R.tensor_calc_reorder_index(fmax, fmin, n_clusters["R5"])
    R.R5_reorder(qlayer.fc1, qlayer.fc2, idx, counts)
    indices["R5"] = idx
    return indices


# ------------------------------------------------------------------------------------------------
# LET initialisation
# ------------------------------------------------------------------------------------------------
def init_let(qlayer, act_scales: Dict[str, torch.Tensor], act_shifts: Dict[str, torch.Tensor], layer_prefix: str,
             cfg: CalibConfig, indices: Optional[dict], dtype, dev, is_llama=False):
    """SmoothQuant-style initial scales act^alpha / w^(1-alpha) and shifts (reference omniquant.py:417-446)."""
    pairs = getattr(qlayer, "LET_PAIRS", {})
    first = next(m for _n, m in qlayer.named_modules() if isinstance(m, QuantLinear) and "q_proj" in _n)
    qlayer.register_parameter("qkt_smooth_scale", nn.Parameter(torch.ones(first.out_features, device=dev, dtype=dtype)))
    for name, module in qlayer.named_modules():
        if not isinstance(module, QuantLinear):
            continue
        for key, (tag, rstage) in pairs.items():
            if key not in name:
                continue
            act = act_scales[f"{layer_prefix}.{name}"].to(device=dev, dtype=dtype).clamp(min=1e-5)
            if indices is not None:
                act = torch.index_select(act, 0, indices[rstage].to(dev))
            weight = module.weight.max(dim=0)[0].clamp(min=1e-5)
            scale = (act.pow(cfg.alpha) / weight.pow(1 - cfg.alpha)).clamp(min=1e-5)
            if cfg.use_shift and not is_llama:
                shift = act_shifts[f"{layer_prefix}.{name}"].to(device=dev, dtype=dtype)
                if indices is not None:
                    shift = torch.index_select(shift, 0, indices[rstage].to(dev))
            else:
                shift = torch.zeros_like(scale)
            qlayer.register_parameter(f"{tag}_smooth_shift", nn.Parameter(shift))
            qlayer.register_parameter(f"{tag}_smooth_scale", nn.Parameter(scale))


# ------------------------------------------------------------------------------------------------
# the per-layer calibration
# ------------------------------------------------------------------------------------------------
def _free_act_stats(qlayer):
    for q in qlayer.act_quantizers():
        q.free()


def _set_eval(qlayer):
    for q in qlayer.act_quantizers():
        q.set_eval_mode()


def _set_dp(qlayer, dp):
    for q in qlayer.act_quantizers():
        q.dp = dp if (dp is not None and dp.active and dp.sync_stats) else None


def _now(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return time.perf_counter()


def calibrate_layer(qlayer, quant_inps, fp_inps, attention_mask, cfg: CalibConfig, act_scales=None, act_shifts=None,
                    layer_prefix: str = "model.decoder.layers.0", dims=None, dp: Optional[DPContext] = None,
                    position_ids=None, logger=None, reorder_stats_layer=None, indices=None, is_llama=False,
                    resume_state=None):
    """Calibrate ONE block in place.

    quant_inps / fp_inps: [n_local, T, H] tensors holding THIS rank's shard of the calibration
    samples (all samples when dp is None; with data parallelism the samples dist.sample_indices(nsamples, rank,
    world), i.e. rank, rank + world, ...); both are overwritten with the block's outputs, as the
    reference does.  ``dims`` = (hidden, ffn, heads) is needed when reordering.  ``resume_state``: the
    layer's entry of a previous run's omni_parameters (reference ``--resume``, omniquant.py:452-453).
    Returns a dict with the per-step losses / gradient norms and the reorder indices.
    """
    dp = dp or DPContext()
    dev = quant_inps.device
    n_local = quant_inps.shape[0]
    n_total = cfg.nsamples
    use_amp = not cfg.deactive_amp
    seconds: Dict[str, float] = {}
    clock = [_now(dev)]

    def lap(name):
        t = _now(dev)
        seconds[name] = seconds.get(name, 0.0) + t - clock[0]
        clock[0] = t

    dtype = torch.float32 if (cfg.deactive_amp and cfg.epochs > 0) else torch.float16
    traincast = (lambda: torch.autocast("cuda", dtype=torch.float16)) if use_amp else nullcontext
    loss_func = MF.mse_loss  # nn.MSELoss(); fp16 operands under autocast take the kernel path
    mask_batch = None
    if attention_mask is not None:
        mask_batch = attention_mask.repeat(cfg.batch_size, 1, 1, 1)
        mask_batch = mask_batch if cfg.deactive_amp else mask_batch.float()

    # ---- reorder ---------------------------------------------------------------------------
    if "0" in cfg.reorder and indices is None:
        enable = {f"R{k}": (str(k) in cfg.reorder) for k in range(1, 6)}
        # the reference gathers the statistics from the ORIGINAL block in fp16 (omniquant.py:376-382);
        # without one, the quantized block is run with quantization switched off
        src = reorder_stats_layer if reorder_stats_layer is not None else qlayer
        handlers = register_hooks(src, *[enable[f"R{k}"] for k in range(1, 6)])
        if src is qlayer:
            qlayer.set_quant_state(False, False)
        with torch.no_grad():
            for j in range(n_local):
                xj = quant_inps[j].unsqueeze(0)
                if reorder_stats_layer is not None:
                    xj = xj.to(torch.float16)
                src(xj, attention_mask=attention_mask)
        for h in handlers:
            h.remove()
        sync_reorder_stats(dp)
        hidden, ffn, heads = dims
        indices = reorder_block(qlayer, hidden, ffn, heads, enable, cfg.n_clusters, dev)

    lap("reorder")
    # ---- full-precision targets -------------------------------------------------------------
    qlayer.set_quant_state(weight_quant=False, act_quant=False)
    pass_amp = (use_amp or cfg.autocast_passes) and dev.type == "cuda"
    if cfg.epochs > 0:
        if not pass_amp:
            with torch.no_grad():
                qlayer.float()  # fp32 path: without autocast the block must match the fp32 activations
        with torch.no_grad(), (torch.autocast("cuda", dtype=torch.float16) if pass_amp else nullcontext()):
            if cfg.aug_loss:  # omniquant.py:405-406
                fp_inps_2 = torch.empty_like(fp_inps)
            for j in range(n_local):
                fp_inps[j] = qlayer(fp_inps[j].unsqueeze(0), attention_mask=attention_mask, position_ids=position_ids)[0]
                if cfg.aug_loss:
                    fp_inps_2[j] = qlayer(quant_inps[j].unsqueeze(0), attention_mask=attention_mask, position_ids=position_ids)[0]

    lap("fp_targets")
    qlayer.set_quant_state(weight_quant=False, act_quant=True)  # weights are quantized by hand before each forward
    qlayer.let = cfg.let
    if cfg.let:
        init_let(qlayer, act_scales, act_shifts, layer_prefix, cfg, indices, dtype, dev, is_llama)
    if resume_state is not None:
        qlayer.load_state_dict({k: v.to(dev) for k, v in resume_state.items()}, strict=False)
    _set_dp(qlayer, dp)

    losses: List[float] = []
    norms: List[float] = []
    graph_steps = 0
    notes: List[str] = []  # decisions a caller without a logger must still be able to see (a graph step given up, say)

    def note(msg):
        notes.append(msg)
        if logger is not None:
            logger.info(msg)
    if cfg.epochs > 0:
        with torch.no_grad():
            qlayer.float()
        groups = []
        if not cfg.weight_mix_quant:
            groups.append({"params": list(qlayer.let_parameters(cfg.use_shift)), "lr": cfg.let_lr})
        groups.append({"params": list(qlayer.lwc_parameters()), "lr": cfg.lwc_lr})
        # (fused AdamW takes GradScaler's found-inf flag on the device: a step without any host synchronisation)
        optimizer = torch.optim.AdamW(groups, weight_decay=cfg.wd,
                                      fused=True if (cfg.fused_adamw and dev.type == "cuda") else None)
        scaler = LossScaler("cuda" if dev.type == "cuda" else "cpu")
        buf = None
        # every rank takes the same number of steps (each one holds a collective): the smallest shard decides
        steps_per_epoch = ((n_total // dp.world) if dp.active else n_local) // cfg.batch_size
        if steps_per_epoch * cfg.batch_size > n_local:
            raise ValueError(f"rank {dp.rank} holds {n_local} samples, fewer than its share of nsamples={n_total} over {dp.world} ranks")
        if dp.active and logger is not None and steps_per_epoch * cfg.batch_size * dp.world != n_total:
            logger.info(f"{n_total - steps_per_epoch * cfg.batch_size * dp.world} samples per epoch are not trained on "
                        f"(nsamples is not a multiple of ranks x batch_size)")
        loss_size_coef = cfg.loss_size_coef
        # per-step loss / gradient norm stay on the device until the end of the epoch
        trace = torch.zeros((2, max(1, steps_per_epoch)), dtype=torch.float32, device=dev)

        def forward_backward(xb, yb, yb2):
            with traincast():
                qlayer.smooth_and_quant_temporary()
                out = qlayer(xb, attention_mask=mask_batch, position_ids=position_ids)[0]
                loss = loss_func(yb, out, kernel=want_graph)
                if cfg.aug_loss:
                    loss = loss + loss_func(yb2, out, kernel=want_graph)
                if cfg.weight_mix_quant:
                    ms, ts = qlayer.get_size_loss()
                    loss = loss + loss_func(ms, ts) * loss_size_coef
            scaler.backward(loss)
            return loss.detach()

        def batch(j):
            i0 = j * cfg.batch_size
            sl = slice(i0, i0 + cfg.batch_size)
            return quant_inps[sl], fp_inps[sl], (fp_inps_2[sl] if cfg.aug_loss else None)

        omni_params = list(qlayer.omni_parameters(cfg.use_shift))  # one walk of the module tree, not one per step
        # A step replayed from a HIP graph: the first step runs eagerly (it warms every lazily built piece of state), then
        # forward + loss + backward are captured ONCE with empty gradients and freed activation statistics, reading the
        # batch from static buffers; every later step copies its batch in, replays, and runs the optimizer eagerly on the
        # gradient tensors the capture created (they are never set to None again: the next replay overwrites them).
        want_graph = (cfg.graph_step and dev.type == "cuda" and not dp.active and not cfg.weight_mix_quant
                      and steps_per_epoch * cfg.epochs > 2)
        graph = static = None
        # the eager first step runs on the stream the capture will use: workspaces and library handles are per stream
        side = torch.cuda.Stream(device=dev) if want_graph else None

        def capture_step():
            xb, yb, yb2 = batch(0)
            st = (torch.empty_like(xb), torch.empty_like(yb), torch.empty_like(yb2) if yb2 is not None else None)
            optimizer.zero_grad(set_to_none=True)
            _free_act_stats(qlayer)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                loss_static = forward_backward(*st)
            _free_act_stats(qlayer)  # (the capture ran the Python side of a forward pass)
            return g, st + (loss_static,)

        def replay_step(j):
            xb, yb, yb2 = batch(j)
            static[0].copy_(xb)
            static[1].copy_(yb)
            if yb2 is not None:
                static[2].copy_(yb2)
            graph.replay()
            return static[3]

        # The first replayed steps, and one step in every cfg.graph_check_every afterwards, are ALSO run eagerly (the
        # optimizer consumes the eager gradients) and compared bit for bit with a replay that follows another replay:
        # PyTorch's multi-block reductions (full sums / means / norms, column sums) do not survive HIP-graph replay on every
        # ROCm build (stale or garbage results when a replay follows a replay; no kernel of this library is involved), and a
        # block variant that still routes one through ATen must fall back to the eager loop rather than train on it.
        verified = [0]
        graph_steps = 0

        def checked_step(j):
            static_grads = [p.grad for p in omni_params]
            for p in omni_params:
                p.grad = None
            loss_e = forward_backward(*batch(j))
            eager_grads = [p.grad for p in omni_params]
            _free_act_stats(qlayer)
            for p, g0 in zip(omni_params, static_grads):
                p.grad = g0
            replay_step(j)
            loss_g = replay_step(j)  # the replay that is compared FOLLOWS a replay, like every later step does
            same = torch.equal(loss_g, loss_e) and all(
                (a is None and b is None) or (a is not None and b is not None and torch.equal(a, b))
                for a, b in zip(static_grads, eager_grads))
            for p, ge in zip(omni_params, eager_grads):
                p.grad = ge  # this step's update uses the eager gradients either way
            return loss_e, same, static_grads

        done = 0
        for epoch in range(cfg.epochs):
            if cfg.weight_mix_quant:
                if epoch == round(cfg.epochs * 0.5):
                    qlayer.set_hard_quant()
                if epoch == round(cfg.epochs * 0.75):
                    qlayer.round_nbits()
                    loss_size_coef = 0
            for j in range(steps_per_epoch):
                if want_graph and done == 1 and graph is None:
                    try:
                        graph, static = capture_step()
                    except RuntimeError as e:  # an operation that cannot be captured (a host synchronisation, say)
                        note(f"{layer_prefix}: the step cannot be captured ({str(e)[:120]}); continuing eagerly")
                        graph = static = None
                        want_graph = False
                        optimizer.zero_grad(set_to_none=True)
                        _free_act_stats(qlayer)
                restore = None
                if graph is not None and (verified[0] < 2 or (cfg.graph_check_every and done % cfg.graph_check_every == 0)):
                    loss, same, restore = checked_step(j)
                    if same:
                        verified[0] += 1
                    elif verified[0] < 2:
                        note(f"{layer_prefix}: the graph-replayed step differs from the eager step at step {done}; "
                             "continuing eagerly")
                        graph = static = restore = None
                        want_graph = False
                    else:
                        raise RuntimeError(f"{layer_prefix}: the graph-replayed step stopped reproducing the eager step at step "
                                           f"{done}; rerun with graph_step=False")
                elif graph is not None:
                    loss = replay_step(j)
                    graph_steps += 1
                elif want_graph:
                    optimizer.zero_grad()
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        loss = forward_backward(*batch(j))
                    torch.cuda.current_stream(dev).wait_stream(side)
                else:
                    optimizer.zero_grad()
                    loss = forward_backward(*batch(j))
                norm, buf = scaler.step(optimizer, omni_params, dp, buf)
                if restore is not None:  # back to the gradient tensors the graph writes
                    for p, g0 in zip(omni_params, restore):
                        p.grad = g0
                trace[0, j] = loss
                trace[1, j] = norm
                _free_act_stats(qlayer)
                done += 1
            # one synchronisation per epoch: the reference tests the loss after every step
            # (omniquant.py:518-520); a non-finite loss only makes GradScaler skip that step, so the
            # delayed test sees the same parameters
            ep_loss, ep_norm = trace[:, :steps_per_epoch].tolist()
            for j, v in enumerate(ep_loss):
                if not math.isfinite(v):
                    raise FloatingPointError(f"loss became {v} at epoch {epoch}, step {j}")
            losses.extend(ep_loss)
            norms.extend(ep_norm)
            if logger is not None:
                k = steps_per_epoch
                logger.info(f"{layer_prefix} epoch {epoch} loss:{np.mean(losses[-k:])} norm:{np.mean(norms[-k:])}")
        qlayer.clear_temp_variable()
        optimizer.zero_grad(set_to_none=True)
        del optimizer
        graph = static = None
    lap("train")

    finalize_layer(qlayer, quant_inps, attention_mask, cfg, dp, position_ids)
    lap("propagate")
    return {"losses": losses, "norms": norms, "indices": indices, "omni_state": qlayer.omni_state_dict(),
            "seconds": seconds, "graph_steps": graph_steps, "steps": len(losses), "notes": notes}


def finalize_layer(qlayer, quant_inps, attention_mask, cfg: CalibConfig, dp: Optional[DPContext] = None, position_ids=None):
    """In-place smoothing + weight quantization with the learned parameters, then the sequential
    propagation pass over all samples (static activation statistics accumulate here, sample by
    sample), scale registration and the switch to eval mode (reference omniquant.py:566-605)."""
    dp = dp or DPContext()
    n_local = quant_inps.shape[0]
    n_total = cfg.nsamples
    use_amp = not cfg.deactive_amp
    if cfg.weight_mix_quant:
        qlayer.round_nbits()
    qlayer.smooth_and_quant_inplace()
    _set_dp(qlayer, None)  # the propagation pass is replicated, not synchronised
    if cfg.epochs <= 0:
        qlayer.half()
    pass_amp = (use_amp or cfg.autocast_passes) and quant_inps.device.type == "cuda"
    cast = (torch.autocast("cuda", dtype=torch.float16) if (pass_amp and cfg.epochs > 0) else nullcontext())
    with torch.no_grad(), cast:
        if dp.active:
            # global sample order 0, 1, 2, ...: sample j lives on rank j % world at local position j // world
            tmp = torch.empty(quant_inps.shape[1:], dtype=quant_inps.dtype, device=quant_inps.device)
            for j in range(n_total):
                own, at = sample_owner(j, dp.world)
                if own == dp.rank:
                    tmp.copy_(quant_inps[at])
                broadcast_from(tmp, own, dp)
                y = qlayer(tmp.unsqueeze(0), attention_mask=attention_mask, position_ids=position_ids)[0]
                if own == dp.rank:
                    quant_inps[at] = y
        else:
            for j in range(n_local):
                quant_inps[j] = qlayer(quant_inps[j].unsqueeze(0), attention_mask=attention_mask, position_ids=position_ids)[0]
    qlayer.register_scales_and_zeros()
    if cfg.epochs > 0:
        qlayer.half()
    _set_eval(qlayer)


def save_omni_parameters(omni_parameters: dict, output_dir: str):
    """``{layer index: OrderedDict(name -> tensor)}`` in the reference's omni_parameters.pth format."""
    os.makedirs(output_dir, exist_ok=True)
    torch.save(omni_parameters, os.path.join(output_dir, "omni_parameters.pth"))


def load_omni_parameters(path: str):
    return torch.load(path, weights_only=True)


def calibrate_blocks(qlayers, inps, attention_mask, cfg: CalibConfig, act_scales=None, act_shifts=None,
                     layer_prefix: str = "model.decoder.layers", dims=None, dp: Optional[DPContext] = None, position_ids=None,
                     logger=None, is_llama=False, output_dir: Optional[str] = None, resume: Optional[dict] = None,
                     offload=None):
    """The layer loop of the reference's ``omniquant()`` (omniquant.py:366-605) over already constructed
    quantized blocks: every block is calibrated against the full-precision stream, both activation
    streams are advanced (the quantized one by the calibrated block, the full-precision one by the
    block with quantization off), and the learned parameters are collected per layer index in the
    format of ``omni_parameters.pth`` (written after every layer when ``output_dir`` is given, like the
    reference does, so an interrupted run can be resumed with ``resume=load_omni_parameters(path)``).

    ``inps``: [n_local, T, H] inputs of the first block held by THIS rank.  ``offload``: optional
    callable(block) invoked after a block is done (e.g. ``lambda b: b.to("cpu")``).
    Returns (omni_parameters dict, per-layer result list, final quantized-stream activations).
    """
    quant_inps = inps
    fp_inps = inps.clone()
    omni_parameters = dict(resume) if resume else {}
    results = []
    for i, qlayer in enumerate(qlayers):
        if logger is not None:
            logger.info(f"=== Start quantize layer {i} ===")
        res = calibrate_layer(qlayer, quant_inps, fp_inps, attention_mask, cfg, act_scales, act_shifts, f"{layer_prefix}.{i}",
                              dims=dims, dp=dp, position_ids=position_ids, logger=logger, is_llama=is_llama,
                              resume_state=(resume or {}).get(i))
        omni_parameters[i] = res["omni_state"]
        results.append(res)
        if output_dir is not None and (dp is None or dp.rank == 0):
            save_omni_parameters(omni_parameters, output_dir)
        if offload is not None:
            offload(qlayer)
    return omni_parameters, results, quant_inps
