"""MI355X-native Florence-2 icon captioner: DaViT vision tower + projector + BART encoder + on-device
greedy decoder, as three static plans (vision+encoder, cross-KV, one decoder step) replayed from HIP
graphs.  Replaces, for `icon_caption`, what the reference reaches through
`processor(...)` + `model.generate(...)` (ref:util/utils.py:116-130) — i.e. transformers' Florence-2
(hf:models/florence2/modeling_florence2.py), BART (hf:models/bart/modeling_bart.py) and the greedy
loop of hf:generation/utils.py:2783-2937.  Host Python only wires pointers; there is no torch math
on the path and no CPU fallback.
"""
import json
import math
import os
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .planner import PlanBuilder, View, torch_dtype

PROMPT_IDS = [0, 2264, 473, 5, 2274, 6190, 116, 2]   # <s>What does the image describe?</s>  
_BUCKETS = tuple(sorted({min(max(int(x), 1), 128) for x in os.environ.get("OMNI_CAPTION_BUCKETS", "8,16,32,64,96,128").split(",")} | {128}))
CLIP_MEAN = (0.485, 0.456, 0.406)
CLIP_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------------------------------ weights
def _legacy_to_native(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Map the remote-code (`trust_remote_code`) Florence-2 key names to the native transformers ones
    (from memory — audited only against the native layout, no legacy checkpoint here)."""
    out = {}
    for k, v in sd.items():
        nk = k
        if nk.startswith("vision_tower."):
            nk = nk.replace(".proj.", ".conv.") if ".convs." in nk else nk
            nk = nk.replace(".fn.dw.", ".")
            nk = nk.replace(".window_attn.norm.", ".norm1.").replace(".channel_attn.norm.", ".norm1.")
            nk = nk.replace(".window_attn.fn.", ".window_attn.").replace(".channel_attn.fn.", ".channel_attn.")
            nk = nk.replace(".ffn.norm.", ".norm2.").replace(".ffn.fn.net.", ".ffn.")
            nk = "model." + nk
        elif nk == "image_projection":
            nk, v = "model.multi_modal_projector.image_projection.weight", v.t().contiguous()
        elif nk.startswith("image_proj_norm."):
            nk = "model.multi_modal_projector." + nk
        elif nk.startswith("image_pos_embed."):
            nk = "model.multi_modal_projector.image_position_embed." + nk[len("image_pos_embed."):]
        elif nk.startswith("visual_temporal_embed."):
            nk = "model.multi_modal_projector." + nk
        elif nk.startswith("language_model.model."):
            nk = "model.language_model." + nk[len("language_model.model."):]
        elif nk == "language_model.lm_head.weight":
            nk = "lm_head.weight"
        elif nk == "language_model.final_logits_bias":
            nk = "final_logits_bias"
        out[nk] = v
    return out


def normalise_config(cfg: dict) -> dict:
    """config.json of the remote-code (`trust_remote_code`) Florence-2 — what the reference actually loads (ref:util/utils.py:63-68,
    weights/icon_caption_florence/config.json) — in the vocabulary of the native transformers config this module reads: the DaViT
    widths are `dim_embed` there (`embed_dim` natively), the learned 2-D position table and the temporal table carry their sizes in
    nested dicts (`image_pos_embed.max_pos_embeddings`, `visual_temporal_embedding.max_temporal_embeddings`), and the generation
    defaults (`no_repeat_ngram_size`, `forced_bos_token_id`, ...) sit in `text_config` or at the top level.  Returns a copy; a native
    config passes through unchanged.  A config that names neither spelling raises a ValueError that says which key is missing —
    never a bare KeyError from deep inside the loader."""
    cfg = json.loads(json.dumps(cfg))
    vc = cfg.get("vision_config")
    tc = cfg.get("text_config")
    if not isinstance(vc, dict) or not isinstance(tc, dict):
        raise ValueError("Florence-2 config.json needs `vision_config` and `text_config` objects")
    if "embed_dim" not in vc and "dim_embed" in vc:
        vc["embed_dim"] = list(vc["dim_embed"])
    pe = vc.get("image_pos_embed")
    if isinstance(pe, dict) and "max_position_embeddings" not in vc and "max_pos_embeddings" in pe:
        vc["max_position_embeddings"] = int(pe["max_pos_embeddings"])
    te = vc.get("visual_temporal_embedding")
    if isinstance(te, dict) and "max_temporal_embeddings" not in vc and "max_temporal_embeddings" in te:
        vc["max_temporal_embeddings"] = int(te["max_temporal_embeddings"])
    need_v = ("embed_dim", "depths", "num_heads", "num_groups", "patch_size", "patch_stride", "patch_padding", "patch_prenorm", "window_size",
              "projection_dim")
    need_t = ("d_model", "vocab_size", "encoder_layers", "decoder_layers", "encoder_attention_heads", "encoder_ffn_dim", "decoder_ffn_dim",
              "max_position_embeddings")
    missing = [f"vision_config.{k}" for k in need_v if k not in vc] + [f"text_config.{k}" for k in need_t if k not in tc]
    if missing:
        raise ValueError("Florence-2 config.json lacks " + ", ".join(missing) + " (native transformers names; the remote-code spellings "
                         "dim_embed / image_pos_embed / visual_temporal_embedding are translated)")
    return cfg


def expected_native_keys(cfg: dict) -> Dict[str, tuple]:
    """Every tensor a Florence-2 checkpoint in the native transformers layout must hold, with its shape, derived from config.json alone
    (hf:models/florence2/modeling_florence2.py module tree).  `FlorenceWeights` audits a checkpoint against it at load:
    a missing, unexpected or mis-shaped tensor is an error that names the offenders — never a silently random layer."""
    vc, tc = cfg["vision_config"], cfg["text_config"]
    dims, depths = vc["embed_dim"], vc["depths"]
    ratio = vc.get("mlp_ratio", 4.0)
    out = {}
    vt = "model.vision_tower."
    cin = vc.get("in_channels", 3)
    for i, C in enumerate(dims):
        k = vc["patch_size"][i]
        out[f"{vt}convs.{i}.conv.weight"] = (C, cin, k, k)
        out[f"{vt}convs.{i}.conv.bias"] = (C,)
        nd = cin if vc["patch_prenorm"][i] else C
        out[f"{vt}convs.{i}.norm.weight"] = out[f"{vt}convs.{i}.norm.bias"] = (nd,)
        hid = int(C * ratio)
        for j in range(depths[i]):
            for blk, attn in (("spatial_block", "window_attn"), ("channel_block", "channel_attn")):
                b = f"{vt}blocks.{i}.{j}.{blk}."
                for cv in ("conv1", "conv2"):
                    out[b + cv + ".weight"] = (C, 1, 3, 3)
                    out[b + cv + ".bias"] = (C,)
                for nm in ("norm1", "norm2"):
                    out[b + nm + ".weight"] = out[b + nm + ".bias"] = (C,)
                out[b + attn + ".qkv.weight"] = (3 * C, C)
                out[b + attn + ".qkv.bias"] = (3 * C,)
                out[b + attn + ".proj.weight"] = (C, C)
                out[b + attn + ".proj.bias"] = (C,)
                out[b + "ffn.fc1.weight"] = (hid, C)
                out[b + "ffn.fc1.bias"] = (hid,)
                out[b + "ffn.fc2.weight"] = (C, hid)
                out[b + "ffn.fc2.bias"] = (C,)
        cin = C
    mp = "model.multi_modal_projector."
    P, D = vc["projection_dim"], tc["d_model"]
    out[mp + "image_projection.weight"] = (P, dims[-1])
    out[mp + "image_proj_norm.weight"] = out[mp + "image_proj_norm.bias"] = (P,)
    out[mp + "image_position_embed.row_embeddings.weight"] = out[mp + "image_position_embed.column_embeddings.weight"] = \
        (vc.get("max_position_embeddings", 50), dims[-1] // 2)
    out[mp + "visual_temporal_embed.pos_idx_to_embed"] = (vc.get("max_temporal_embeddings", 100), dims[-1])
    lm = "model.language_model."
    out[lm + "shared.weight"] = (tc["vocab_size"], D)
    for side, n_layers, ffn in (("encoder", tc["encoder_layers"], tc["encoder_ffn_dim"]), ("decoder", tc["decoder_layers"], tc["decoder_ffn_dim"])):
        out[f"{lm}{side}.embed_positions.weight"] = (tc["max_position_embeddings"] + 2, D)
        out[f"{lm}{side}.layernorm_embedding.weight"] = out[f"{lm}{side}.layernorm_embedding.bias"] = (D,)
        for l in range(n_layers):
            b = f"{lm}{side}.layers.{l}."
            attns = ("self_attn",) if side == "encoder" else ("self_attn", "encoder_attn")
            for a in attns:
                for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    out[b + f"{a}.{pj}.weight"] = (D, D)
                    out[b + f"{a}.{pj}.bias"] = (D,)
                out[b + f"{a}_layer_norm.weight"] = out[b + f"{a}_layer_norm.bias"] = (D,)
            out[b + "fc1.weight"] = (ffn, D); out[b + "fc1.bias"] = (ffn,)
            out[b + "fc2.weight"] = (D, ffn); out[b + "fc2.bias"] = (D,)
            out[b + "final_layer_norm.weight"] = out[b + "final_layer_norm.bias"] = (D,)
    return out


# tensors a checkpoint MAY carry beyond the list above: the tied output embedding, the (all-zero) logits bias, the encoder / decoder
# copies of the shared embedding that older exports duplicate
_OPTIONAL_KEYS = {"lm_head.weight": lambda cfg: (cfg["text_config"]["vocab_size"], cfg["text_config"]["d_model"]),
                  "final_logits_bias": lambda cfg: (1, cfg["text_config"]["vocab_size"]),
                  "model.language_model.encoder.embed_tokens.weight": lambda cfg: (cfg["text_config"]["vocab_size"], cfg["text_config"]["d_model"]),
                  "model.language_model.decoder.embed_tokens.weight": lambda cfg: (cfg["text_config"]["vocab_size"], cfg["text_config"]["d_model"])}


def audit_checkpoint(sd: Dict[str, torch.Tensor], cfg: dict) -> None:
    """Strict audit of a (native-layout) state dict against config.json: raises ValueError naming what is wrong."""
    want = expected_native_keys(cfg)
    missing = sorted(k for k in want if k not in sd)
    unexpected = sorted(k for k in sd if k not in want and k not in _OPTIONAL_KEYS)
    shapes = sorted(f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k]))
    shapes += sorted(f"{k}: {tuple(sd[k].shape)} != {_OPTIONAL_KEYS[k](cfg)}" for k in _OPTIONAL_KEYS
                     if k in sd and tuple(sd[k].shape) != tuple(_OPTIONAL_KEYS[k](cfg)))
    if missing or unexpected or shapes:
        show = lambda l: ", ".join(l[:6]) + (f" ... (+{len(l) - 6})" if len(l) > 6 else "")
        raise ValueError("Florence-2 checkpoint does not match its config.json: "
                         + "; ".join(p for p in (f"{len(missing)} missing [{show(missing)}]" if missing else "",
                                                 f"{len(unexpected)} unexpected [{show(unexpected)}]" if unexpected else "",
                                                 f"{len(shapes)} mis-shaped [{show(shapes)}]" if shapes else "") if p))


class FlorenceWeights:
    def __init__(self, model_dir):
        from safetensors.torch import load_file
        d = Path(model_dir)
        self.dir = d
        self.cfg = normalise_config(json.loads((d / "config.json").read_text()))
        gpath = d / "generation_config.json"
        self.gen = json.loads(gpath.read_text()) if gpath.exists() else {}
        files = sorted(d.glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors under {d}")
        sd = {}
        for f in files:
            sd.update(load_file(str(f)))
        if any(k.startswith("vision_tower.") for k in sd):
            sd = _legacy_to_native(sd)
        audit_checkpoint(sd, self.cfg)           # strict: every tensor the config implies, with its shape, nothing unknown
        self.sd = {k: v.float() for k, v in sd.items()}
        if "lm_head.weight" not in self.sd:
            self.sd["lm_head.weight"] = self.sd["model.language_model.shared.weight"]
        vc, tc = self.cfg["vision_config"], self.cfg["text_config"]
        self.embed_dim = vc["embed_dim"]; self.depths = vc["depths"]; self.heads = vc["num_heads"]
        self.groups = vc["num_groups"]; self.patch = list(zip(vc["patch_size"], vc["patch_stride"], vc["patch_padding"]))
        self.prenorm = vc["patch_prenorm"]; self.window = vc["window_size"]
        self.d_model = tc["d_model"]; self.n_heads = tc["encoder_attention_heads"]
        self.enc_layers, self.dec_layers = tc["encoder_layers"], tc["decoder_layers"]
        self.vocab = self.sd["lm_head.weight"].shape[0]
        self.embed_scale = math.sqrt(self.d_model) if tc.get("scale_embedding", False) else 1.0
        # generation defaults: generation_config.json, else text_config, else the top level of config.json (where the remote-code config and
        # `_from_model_config` exports keep them), else the transformers default (no n-gram ban, nothing forced)
        g = lambda k, dflt: self.gen.get(k, tc.get(k, self.cfg.get(k, dflt)))
        self.pad, self.bos, self.eos = g("pad_token_id", 1), g("bos_token_id", 0), g("eos_token_id", 2)
        self.start = g("decoder_start_token_id", 2)
        self.ngram = g("no_repeat_ngram_size", 0) or 0
        fb, fe = g("forced_bos_token_id", None), g("forced_eos_token_id", None)
        self.forced_bos = -1 if fb is None else fb
        self.forced_eos = -1 if fe is None else fe
        assert self.window == 12 and self.d_model // self.n_heads == 64


_DECODE_TUNING = "unset"


def decode_tuning(device=None):
    """The committed tuning table of the decode step's GEMMs (`omniparser_amd/decode_tuning_gfx950.json`, written by
    tools/decode_autotune.py on the MI355X): {shape key: (tile code, split-K count)} for the row counts a merged decode plan can have.
    A decode step is launch-bound (~110 kernels of 5-15 us); the launcher's throughput heuristic cuts its GEMMs into 6 K splits of four
    slices plus a reduce launch where fewer, longer blocks are faster.  Tuning changes the order in which K partials are summed, nothing
    else (the logits' last bits already depend on the row count, see _DecodePlans); OMNI_DECODE_TUNING=0 turns it off; a device that is
    not gfx950, or a missing file, means the heuristic.  NO TABLE IS COMMITTED: measured in round 6 (profiles/r6_s13_decode_autotune.json),
    the best choices beat the heuristic by 2.8 % of a step graph at 128 rows, 0.3-0.5 % at 160-320 rows and nothing at 352 / 384 rows
    (the benched load) — the heuristic stays; the tool and this loader remain for other chips / row counts."""
    global _DECODE_TUNING
    if device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available():
        if not getattr(torch.cuda.get_device_properties(device), "gcnArchName", "").startswith("gfx950"):
            return None
    if _DECODE_TUNING == "unset":
        path = Path(__file__).resolve().parent / "decode_tuning_gfx950.json"
        _DECODE_TUNING = None
        if os.environ.get("OMNI_DECODE_TUNING", "1") != "0" and path.exists():
            _DECODE_TUNING = {k: tuple(v) for k, v in json.loads(path.read_text())["choices"].items()}
    return _DECODE_TUNING


# ------------------------------------------------------------------------------------------ plans
class _StepPlans:
    """The decoder-step plan over B rows (embedding, 6 BART decoder layers with self-KV cache and fixed cross-KV, lm_head, logits
    processors + arg-max on the device) and its state: shared by _CaptionPlans (encode + decode of one micro-batch) and _DecodePlans."""

    def _build_step(self, cap, B, max_new, S, cross_kv, ws=None):
        w, dev, dt = cap.w, cap.device, cap.dtype
        sd, wc = w.sd, cap._wcache
        D, nh, lm = w.d_model, w.n_heads, "model.language_model."
        self.T = max_new + 1
        pb = PlanBuilder(dev, dt)                          # weights / tables missing from the model's cache are uploaded through it

        def packed(key, make):
            ck = (key, dt)
            if ck not in wc:
                wt, b = make()
                wc[ck] = (pb.pack_weight(wt if wt.dim() == 4 else wt[:, :, None, None]), pb.upload(b.float()) if b is not None else None)
            return wc[ck]

        def f32(key):
            ck = (key, "f32")
            if ck not in wc:
                wc[ck] = pb.upload(sd[key].float())
            return wc[ck]

        pd_ = PlanBuilder(dev, dt)
        pd_.ws = ws
        if isinstance(self, _DecodePlans):
            pd_.conv_tuning = decode_tuning(dev)          # merged decode plans only: the per-micro-batch step plans keep the heuristic
        self.pd = pd_
        T = self.T
        self.B = B
        self.ids = pd_.raw((B, T), torch.int32)
        self.finished = pd_.raw((B,), torch.int32)
        self.step = pd_.raw((1,), torch.int32)
        esz = 4 if dt == L.F32 else 2

        def dlinear(key, xin: View, out: View, act=L.ACT_NONE, res=None, keys=None, bias=True):
            keys = keys or [key]
            def make():
                wt = torch.cat([sd[k + ".weight"] for k in keys], 0)
                b = torch.cat([sd[k + ".bias"] for k in keys], 0) if bias else None
                return wt, b
            wp, bp = packed("|".join(keys), make)
            return pd_.conv(xin, wp, bp, out, 1, act=act, res=res)

        def dln(key, xin: View, out: View):
            pd_.add_op(L.make_op(L.OP_LAYERNORM, dt, p=[xin.ptr, None, f32(key + ".weight").data_ptr(),
                                                       f32(key + ".bias").data_ptr(), out.ptr],
                                 i={0: B, 1: 1, 3: D, 5: 0}, f={0: 1e-5}))
            return out

        ck = ("dectab", dt)
        if ck not in wc:
            wc[ck] = (pb.upload(sd[lm + "shared.weight"].to(torch_dtype(dt))),
                      pb.upload(sd[lm + "decoder.embed_positions.weight"].to(torch_dtype(dt))))
        table, dpos = wc[ck]
        pd_.keep += [table, dpos]
        e = pd_.alloc(B, 1, 1, D)
        pd_.add_op(L.make_op(L.OP_EMBED_STEP, dt, p=[table.data_ptr(), dpos.data_ptr(), self.ids.data_ptr(), None, e.ptr,
                                                     None, self.step.data_ptr()],
                             i={0: B, 3: D, 4: T, 5: 2}, f={0: w.embed_scale}))
        xd = pd_.alloc(B, 1, 1, D)
        dln(lm + "decoder.layernorm_embedding", e, xd)
        dqkv = pd_.alloc(B, 1, 1, 3 * D)
        dq = pd_.alloc(B, 1, 1, D)
        da = pd_.alloc(B, 1, 1, D)
        dt_ = pd_.alloc(B, 1, 1, D)
        dffn = pd_.alloc(B, 1, 1, sd[lm + "decoder.layers.0.fc1.weight"].shape[0])
        self.self_k = [pd_.alloc(B, T, 1, D) for _ in range(w.dec_layers)]
        self.self_v = [pd_.alloc(B, T, 1, D) for _ in range(w.dec_layers)]
        for l in range(w.dec_layers):
            pre = f"{lm}decoder.layers.{l}."
            dlinear(None, xd, dqkv, keys=[pre + "self_attn.q_proj", pre + "self_attn.k_proj", pre + "self_attn.v_proj"])
            pd_.add_op(L.make_op(L.OP_ATTN_DECODE, dt,
                                 p=[dqkv.ptr, dqkv.ptr, dqkv.ptr, self.self_k[l].ptr, da.ptr, self.self_v[l].ptr, self.step.data_ptr()],
                                 i={0: 3 * D, 1: 0, 2: 3 * D, 3: D, 4: 2 * D, 5: D, 6: nh, 7: 0, 8: T, 9: D, 10: B, 11: D},
                                 f={0: 64 ** -0.5}))
            dlinear(pre + "self_attn.out_proj", da, dt_, res=xd)
            dln(pre + "self_attn_layer_norm", dt_, xd)
            dlinear(pre + "encoder_attn.q_proj", xd, dq)
            kv = self.cross_kv[l]
            pd_.add_op(L.make_op(L.OP_ATTN_DECODE, dt,
                                 p=[dq.ptr, None, None, kv.ptr, da.ptr, kv.ptr + D * esz, None],
                                 i={0: D, 1: 0, 2: 0, 3: 0, 4: 0, 5: D, 6: nh, 7: S, 8: S, 9: D, 10: B, 11: 2 * D},
                                 f={0: 64 ** -0.5}))
            dlinear(pre + "encoder_attn.out_proj", da, dt_, res=xd)
            dln(pre + "encoder_attn_layer_norm", dt_, xd)
            dlinear(pre + "fc1", xd, dffn, act=L.ACT_GELU)
            dlinear(pre + "fc2", dffn, dt_, res=xd)
            dln(pre + "final_layer_norm", dt_, xd)
        logits = pd_.alloc(B, 1, 1, w.vocab)
        self.logits = logits
        wp, _ = packed("lm_head", lambda: (sd["lm_head.weight"], None))
        pd_.conv(xd, wp, None, logits, 1)
        flb = None
        if "final_logits_bias" in sd:
            flb = f32("final_logits_bias")
            pd_.keep.append(flb)
        pd_.add_op(L.make_op(L.OP_GREEDY_STEP, dt,
                             p=[logits.ptr, flb.data_ptr() if flb is not None else None, self.ids.data_ptr(),
                                self.finished.data_ptr(), None, None, self.step.data_ptr()],
                             i={0: B, 1: w.vocab, 2: w.vocab, 3: T, 4: max_new, 5: w.ngram, 6: w.bos, 7: w.eos, 8: w.pad,
                                9: w.forced_bos, 10: w.forced_eos, 11: 1}))
        self.step_flops = pd_.flops
        self.step_plan = pd_.build()
        self.start_token = w.start

    def reset(self):
        self.ids.zero_()
        self.ids[:, 0] = self.start_token
        self.finished.zero_()
        self.step.zero_()


class _DecodePlans(_StepPlans):
    """Decode for the crops of SEVERAL caption micro-batches at once.  The encode side of a batch of screenshots runs in micro-batches
    of <= 128 crops (~25 GB of activations each at 768x768 with activation reuse, 60 GB without); the decode side is 20 steps of ~110 small kernels whose cost hardly
    depends on the row count (launch-bound GEMMs over 128 rows) — per micro-batch that was 29 ms, 80 ms of a 750 ms step.  Here the
    cross-attention K / V of every micro-batch are copied (6 x 460 MB per 128 crops, ~1 ms) into the row range of ONE decode plan over
    B rows, which then runs its 20 steps once.  Rows are independent: ids per crop are what the per-micro-batch decode produces
    (same kernels; the split-K choice of the step GEMMs, hence the last bits of the logits, depends on the row count)."""

    def __init__(self, cap: "Florence2Captioner", B: int, R: int, max_new: int):
        w, dev, dt = cap.w, cap.device, cap.dtype
        self.B, self.R = B, R
        S = (R // 32) ** 2 + 1 + len(PROMPT_IDS)
        self.S = S
        pk = PlanBuilder(dev, dt)
        # zero-initialised: the rows behind the last crop are never written (the micro-batches copy their own rows only) but ARE decoded;
        # recycled allocator memory there can hold NaN bit patterns, whose logits are all-NaN rows (found by the one-process GPU suite)
        self.cross_kv = [pk.alloc(B, S, 1, 2 * w.d_model, zero=True) for _ in range(w.dec_layers)]
        self._keep = pk.keep
        self._build_step(cap, B, max_new, S, self.cross_kv)
        self.free_evt = None       # recorded behind the decode that last used this plan on another stream (pipelined batches)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if cap.use_graph:
            self.reset()
            self.step_plan.run(cap.stream)
            cap.stream.synchronize()
            self.step_plan.capture(cap.stream)
            cap.stream.synchronize()


class _CaptionPlans(_StepPlans):
    """Static plans for B crops at resolution R.

    arena: another _CaptionPlans of the same resolution and a capacity >= B whose BUFFERS this one uses (round 6).  Such a plan set is
    encode-only: the same op list over the first B rows of every tensor of `arena` (PlanBuilder._from_arena), captured as its own
    hipGraph — the last micro-batch of a caption batch then computes exactly its own rows instead of a padded bucket (89 rows instead
    of 96 at the benched load), without a second set of activations (~19 GB at 96 rows).  Built on demand by `encode_rows`."""

    def __init__(self, cap: "Florence2Captioner", B: int, R: int, max_new: int, arena: Optional["_CaptionPlans"] = None, stream=None):
        w, dev, dt = cap.w, cap.device, cap.dtype
        sd = w.sd
        self.B, self.R, self.T = B, R, max_new + 1
        pb = PlanBuilder(dev, dt)
        pb.reuse = bool(cap.reuse_activations)
        if arena is not None:
            assert arena.R == R and arena.B >= B and arena.arena is None
            pb.arena = iter(arena.pb.alloc_log)
        self.arena = arena
        self._row_plans = {}      # n -> encode-only plan set of exactly n rows in THIS plan set's buffers (encode_rows), LRU
        self.pb = pb
        V = pb.V
        wc = cap._wcache

        # f32 plans run the encode-side linear layers on the pre-split LDS-DMA GEMM (csrc/gemm_dma.hip): weights in format B,
        # inputs written pre-split by their producers.  OMNI_GEMM_DMA=0 keeps every layer on the register-staged kernels.
        use_dma = dt == L.F32 and pb.split and os.environ.get("OMNI_GEMM_DMA", "1") != "0"
        self.use_dma = use_dma
        # format B is decided per producer / consumer GROUP, not per layer: a DaViT stage (or the encoder) uses it only when EVERY
        # linear layer fed by its LayerNorm / GELU / attention outputs takes the LDS-DMA GEMM (N % 128 == 0, K % 32 == 0), i.e. when
        # the width is a multiple of 128 — true for all Florence-2 widths (128..1024, 768); narrower stand-ins stay on f32 tensors
        grp = {"dma": use_dma}

        def packed(key, make, dma=False):
            ck = (key, dt, "dma") if dma else (key, dt)
            if ck not in wc:
                wt, b = make()
                wc[ck] = (pb.pack_weight_dma(wt) if dma else pb.pack_weight(wt if wt.dim() == 4 else wt[:, :, None, None]),
                          pb.upload(b.float()) if b is not None else None)
            return wc[ck]

        def f32(key):
            ck = (key, "f32")
            if ck not in wc:
                wc[ck] = pb.upload(sd[key].float())
            return wc[ck]

        def tokens(v: View) -> View:
            return View(v.t.view(v.B, v.H * v.W, 1, v.ld), v.coff, v.C, v.fmt)

        def linear(key, x: View, out: View, act=L.ACT_NONE, res=None, bias=True, keys=None, out_split=False):
            """nn.Linear.  With the LDS-DMA GEMM an f32 input is converted to format B IN PLACE first (callers only pass
            tensors whose f32 content nobody else reads); out_split: the epilogue writes format B for the next GEMM."""
            keys = keys or [key]
            def make():
                wt = torch.cat([sd[k + ".weight"] for k in keys], 0)
                b = torch.cat([sd[k + ".bias"] for k in keys], 0) if bias else None
                return wt, b
            n_out, k_in = out.C, x.C
            dma = grp["dma"] and n_out % 128 == 0 and k_in % 32 == 0 and x.ld % 16 == 0 and x.coff % 16 == 0
            assert dma or x.fmt != "split", f"{keys}: split input for a layer that cannot take the LDS-DMA GEMM"
            wp, bp = packed("|".join(keys), make, dma=dma)
            if dma and x.fmt != "split":
                pb.split_convert(x)
            xt, ot = tokens(x), tokens(out)
            pb.conv(xt, wp, bp, ot, 1, act=act, res=tokens(res) if res is not None else None, out_split=dma and out_split)
            out.fmt = ot.fmt
            return out

        def layernorm(key, x: View, out: View, add=None, period=0, eps=1e-5, split=None):
            """split=None: f32 output; split=out: write format B instead (the only consumer is an LDS-DMA GEMM);
            split=<other View>: f32 into `out` AND format B into that view (post-LN rows that are also a residual)."""
            rows = x.B * x.H * x.W
            omode, y2 = 0, None
            if grp["dma"] and split is not None and x.C % 16 == 0:
                omode, y2 = (1, None) if split is out else (2, split)
            pb.add_op(L.make_op(L.OP_LAYERNORM, dt, p=[x.ptr, add.data_ptr() if add is not None else None,
                                                      f32(key + ".weight").data_ptr(), f32(key + ".bias").data_ptr(), out.ptr,
                                                      y2.ptr if y2 is not None else None],
                                i={0: rows, 1: 1, 3: x.C, 5: period, 6: omode}, f={0: eps}))
            out.fmt = "split" if omode == 1 else "f32"
            if y2 is not None:
                y2.fmt = "split"
            return out

        def dwconv(key, x: View, out: View):
            ck = (key, dt)
            if ck not in wc:
                wt = sd[key + ".weight"]                      # [C,1,3,3] -> [3][3][C]
                wc[ck] = (pb.upload(wt[:, 0].permute(1, 2, 0).contiguous().to(torch_dtype(dt))), pb.upload(sd[key + ".bias"].float()))
            wp, bp = wc[ck]
            pb.keep += [wp, bp]
            pb.add_op(L.make_op(L.OP_DWCONV3, dt, p=[x.ptr, wp.data_ptr(), bp.data_ptr(), None, out.ptr],
                                i={0: x.B, 1: x.H, 2: x.W, 3: x.C}))
            return out

        # x + dwconv(x) and the LayerNorm behind it as ONE strip kernel (csrc/caption_ops.hip::dwln_strip_kernel) wherever it exists
        # (f32 plans, C = 128 / 256 / 512: DaViT stages 0-2); Florence2Captioner.fuse_dwln = False keeps the two separate kernels
        fuse_env = cap.fuse_dwln
        # the attention kernels write their output pre-split for the projection GEMM (no split_convert pass: 22.5 -> 0.3 ms per step on
        # the MI355X, BENCH_r02 extra.ab_opt_in_kernels); Florence2Captioner.attn_split_out = False = f32 output + in-place conversion
        attn_split = use_dma and cap.attn_split_out

        def dwconv_ln(conv_key, norm_key, x: View, y1: View, hout: View):
            """x1 = x + dwconv(x); h = LN(x1) — one kernel (the conv result never leaves registers before the statistics)."""
            if not (fuse_env and dt == L.F32 and x.C in (128, 256, 512)):
                dwconv(conv_key, x, y1)
                return layernorm(norm_key, y1, hout, split=hout)      # hbuf only feeds qkv / fc1
            ck = (conv_key, dt)
            if ck not in wc:
                wt = sd[conv_key + ".weight"]
                wc[ck] = (pb.upload(wt[:, 0].permute(1, 2, 0).contiguous().to(torch_dtype(dt))), pb.upload(sd[conv_key + ".bias"].float()))
            wp, bp = wc[ck]
            pb.keep += [wp, bp]
            pb.add_op(L.make_op(L.OP_DWCONV3_LN, dt,
                                p=[x.ptr, wp.data_ptr(), bp.data_ptr(), hout.ptr, y1.ptr,
                                   f32(norm_key + ".weight").data_ptr(), f32(norm_key + ".bias").data_ptr()],
                                i={0: x.B, 1: x.H, 2: x.W, 3: x.C, 6: 1 if grp["dma"] else 0}, f={0: 1e-5}))
            hout.fmt = "split" if grp["dma"] else "f32"
            return hout

        # ---------------- input + vision tower
        self.x_in = pb.alloc(B, R, R, V, zero=True)
        x = self.x_in
        vt = "model.vision_tower."
        self.chan_ws = None
        self.stage_out = []          # output of every DaViT stage (valid after the encode plan unless cap.reuse_activations: each stage owns its buffers)
        for s in range(4):
            C = w.embed_dim[s]
            grp["dma"] = use_dma and C % 128 == 0
            k, st, pd = w.patch[s]
            Ho = (x.H + 2 * pd - k) // st + 1
            conv_key = f"{vt}convs.{s}.conv"
            if w.prenorm[s]:
                xn = pb.alloc(B, x.H, x.W, x.C)
                layernorm(f"{vt}convs.{s}.norm", x, xn)
                cur = pb.alloc(B, Ho, Ho, C)
                wp, bp = packed(conv_key, lambda ck=conv_key: (sd[ck + ".weight"], sd[ck + ".bias"]))
                pb.conv(xn, wp, bp, cur, k, st, pd)
                pb.release(xn, *([] if x is self.x_in else [x]))   # reuse_activations: the previous stage's output has had its last reader
            else:
                t0 = pb.alloc(B, Ho, Ho, C)
                if cap.patch_rows and dt == L.F32 and pb.split and V == 4 and k <= 8 and x is self.x_in:
                    # the 7 x 7 / stride-4 patch embedding over the 3 (stored: 4) input channels as a 7 x 1 convolution over 8 consecutive
                    # pixels on the split-f16 MFMA kernel (PlanBuilder.conv_patch) instead of K = 196 on the exact-f32 one (81 TF/s)
                    ck2 = (conv_key, dt, "patch")
                    if ck2 not in wc:
                        wc[ck2] = (pb.pack_weight_patch(sd[conv_key + ".weight"], V), pb.upload(sd[conv_key + ".bias"].float()))
                    wp, bp = wc[ck2]
                    pb.conv_patch(x, wp, bp, t0, k, st, pd)
                else:
                    ck2 = (conv_key, dt, "pad")
                    if ck2 not in wc:
                        wc[ck2] = (pb.pack_weight(sd[conv_key + ".weight"], cin_pad=V), pb.upload(sd[conv_key + ".bias"].float()))
                    wp, bp = wc[ck2]
                    pb.conv(x, wp, bp, t0, k, st, pd)
                cur = pb.alloc(B, Ho, Ho, C)
                layernorm(f"{vt}convs.{s}.norm", t0, cur)
                pb.release(t0)               # (x is the plan's input here: the crop kernels write it, never released)
            H = Ho
            N = H * H
            A_, B_ = cur, pb.alloc(B, H, H, C)
            hbuf = pb.alloc(B, H, H, C)
            qkv = pb.alloc(B, H, H, 3 * C)
            att = pb.alloc(B, H, H, C)
            # the FFN of the C = 128 stage is one kernel whose hidden activations stay in registers (csrc/gemm_dma.hip::mlp_fused_kernel):
            # no [B, H, H, 4C] tensor (9.7 GB of a 128-crop plan at 768x768), 12 instead of 44 bytes of HBM traffic per token-channel
            mlp_one = grp["dma"] and cap.fuse_mlp and C == 128 and sd[f"{vt}blocks.{s}.0.spatial_block.ffn.fc1.weight"].shape[0] == 512
            ffn = None if mlp_one else pb.alloc(B, H, H, 4 * C)
            chunk_tokens = 1024
            chunks = (N + chunk_tokens - 1) // chunk_tokens
            cws = pb.raw((B * w.groups[s] * chunks * 1024,), torch.float32, zero=False)
            for blk in range(w.depths[s]):
                for kind in ("spatial_block", "channel_block"):
                    pre = f"{vt}blocks.{s}.{blk}.{kind}."
                    dwconv_ln(pre + "conv1", pre + "norm1", A_, B_, hbuf)
                    if kind == "spatial_block":
                        linear(pre + "window_attn.qkv", hbuf, qkv)
                        qb = f32(pre + "window_attn.qkv.bias")
                        nw = ((H + 11) // 12) ** 2
                        pb.add_op(L.make_op(
                            L.OP_ATTN_ROWS, dt,
                            p=[qkv.ptr, qkv.ptr, qkv.ptr, None, att.ptr, qb.data_ptr() + 4 * C, qb.data_ptr() + 8 * C],
                            i={0: 3 * C, 1: 3 * C, 2: 3 * C, 3: C, 4: 0, 5: C, 6: 2 * C, 7: 0, 8: w.heads[s], 9: 144, 10: 144,
                               11: B * nw, 12: 1, 13: H, 14: H, 15: C // w.heads[s], 16: 1 if (attn_split and grp["dma"]) else 0},
                            f={0: (C // w.heads[s]) ** -0.5}))
                        att.fmt = "split" if (attn_split and grp["dma"]) else "f32"
                        linear(pre + "window_attn.proj", att, B_, res=B_)
                    else:
                        linear(pre + "channel_attn.qkv", hbuf, qkv)
                        pb.add_op(L.make_op(L.OP_CHAN_ATTN, dt, p=[qkv.ptr, None, None, None, att.ptr, cws.data_ptr()],
                                            i={0: B, 1: N, 3: C, 4: w.groups[s], 5: chunk_tokens, 6: 1 if (attn_split and grp["dma"]) else 0}))
                        att.fmt = "split" if (attn_split and grp["dma"]) else "f32"
                        linear(pre + "channel_attn.proj", att, B_, res=B_)
                    dwconv_ln(pre + "conv2", pre + "norm2", B_, A_, hbuf)
                    if mlp_one:
                        k1, k2 = pre + "ffn.fc1", pre + "ffn.fc2"
                        w1p, b1p = packed(k1, lambda k=k1: (sd[k + ".weight"], sd[k + ".bias"]), dma=True)
                        ck = (k2, dt, "dma-kperm")
                        if ck not in wc:
                            wc[ck] = (pb.pack_weight_dma(sd[k2 + ".weight"], kperm=True), pb.upload(sd[k2 + ".bias"].float()))
                        w2p, b2p = wc[ck]
                        pb.mlp_fused(tokens(hbuf), w1p, b1p, w2p, b2p, tokens(A_), tokens(A_))
                        continue
                    linear(pre + "ffn.fc1", hbuf, ffn, act=L.ACT_GELU, out_split=True)
                    linear(pre + "ffn.fc2", ffn, A_, res=A_)
            x = A_
            self.stage_out.append(A_)
            # reuse_activations: the stage's scratch tensors back the (smaller) tensors of the next stages
            pb.release(B_, hbuf, qkv, att, ffn, cws)
        self.vision_out = x
        grp["dma"] = use_dma
        # ---------------- projector
        mp = "model.multi_modal_projector."
        h = x.H
        Cv = x.C
        ck = ("pos2d", h)
        if ck not in wc:
            col = sd[mp + "image_position_embed.column_embeddings.weight"][:h]
            row = sd[mp + "image_position_embed.row_embeddings.weight"][:h]
            pos = torch.cat([col.unsqueeze(0).repeat(h, 1, 1), row.unsqueeze(1).repeat(1, h, 1)], -1).reshape(h * h, Cv)
            wc[ck] = (pb.upload(pos.float()), pb.upload(sd[mp + "visual_temporal_embed.pos_idx_to_embed"][0].float()))
        pos2d, temporal = wc[ck]
        pb.keep += [pos2d, temporal]
        n_img = h * h + 1
        self.n_img = n_img
        pp = pb.alloc(B, n_img, 1, Cv)
        pb.add_op(L.make_op(L.OP_PROJ_PREP, dt, p=[x.ptr, pos2d.data_ptr(), temporal.data_ptr(), None, pp.ptr],
                            i={0: B, 1: h * h, 3: Cv}))
        D = w.d_model
        pj = pb.alloc(B, n_img, 1, D)
        linear(mp + "image_projection", pp, pj, bias=False)
        img_feat = pb.alloc(B, n_img, 1, D)
        layernorm(mp + "image_proj_norm", pj, img_feat)
        self.img_feat = img_feat
        # ---------------- encoder
        lm = "model.language_model."
        S = n_img + len(PROMPT_IDS)
        self.S = S
        ck = ("prompt", dt)
        if ck not in wc:
            emb = sd[lm + "shared.weight"][torch.tensor(PROMPT_IDS)] * w.embed_scale
            wc[ck] = pb.upload(emb.to(torch_dtype(dt)))
        txt = wc[ck]
        pb.keep.append(txt)
        enc = pb.alloc(B, S, 1, D)
        pb.add_op(L.make_op(L.OP_ASSEMBLE, dt, p=[img_feat.ptr, txt.data_ptr(), None, None, enc.ptr],
                            i={0: B, 1: n_img, 2: len(PROMPT_IDS), 3: D}))
        ck = ("encpos", S, dt)
        if ck not in wc:
            wc[ck] = pb.upload(sd[lm + "encoder.embed_positions.weight"][2:2 + S].to(torch_dtype(dt)))
        encpos = wc[ck]
        pb.keep.append(encpos)
        xa = pb.alloc(B, S, 1, D)
        dma_enc = grp["dma"] = use_dma and D % 128 == 0     # every encoder linear (K = D or 4D, N multiple of D) then takes the DMA GEMM
        xs = pb.alloc(B, S, 1, D) if dma_enc else None        # format-B twin of xa (xa itself stays f32: it is the residual)
        layernorm(lm + "encoder.layernorm_embedding", enc, xa, add=encpos, period=S, split=xs)
        xin = xs if dma_enc else xa
        qkv = pb.alloc(B, S, 1, 3 * D)
        att = pb.alloc(B, S, 1, D)
        tmp = pb.alloc(B, S, 1, D)
        ffn = pb.alloc(B, S, 1, sd[lm + "encoder.layers.0.fc1.weight"].shape[0])
        nh = w.n_heads
        for l in range(w.enc_layers):
            pre = f"{lm}encoder.layers.{l}."
            linear(None, xin, qkv, keys=[pre + "self_attn.q_proj", pre + "self_attn.k_proj", pre + "self_attn.v_proj"])
            pb.add_op(L.make_op(L.OP_ATTN_ROWS, dt, p=[qkv.ptr, qkv.ptr, qkv.ptr, None, att.ptr],
                                i={0: 3 * D, 1: 3 * D, 2: 3 * D, 3: D, 4: 0, 5: D, 6: 2 * D, 7: 0, 8: nh, 9: S, 10: S, 11: B,
                                   12: 0, 15: 64, 16: 1 if (attn_split and dma_enc) else 0}, f={0: 64 ** -0.5}))
            att.fmt = "split" if (attn_split and dma_enc) else "f32"
            linear(pre + "self_attn.out_proj", att, tmp, res=xa)
            layernorm(pre + "self_attn_layer_norm", tmp, xa, split=xs)
            linear(pre + "fc1", xin, ffn, act=L.ACT_GELU, out_split=True)
            linear(pre + "fc2", ffn, tmp, res=xa)
            layernorm(pre + "final_layer_norm", tmp, xa, split=xs)
        self.enc_out = xa
        # ---------------- cross-attention K/V of every decoder layer (computed once per batch)
        self.cross_kv = []
        for l in range(w.dec_layers):
            pre = f"{lm}decoder.layers.{l}.encoder_attn."
            kv = pb.alloc(B, S, 1, 2 * D)
            linear(None, xin, kv, keys=[pre + "k_proj", pre + "v_proj"])
            self.cross_kv.append(kv)
        self.free_evt = None        # recorded behind the last use of this plan's buffers when that was on another stream (parse_stream)
        self.n_encode_ops = len(pb.ops)
        self.encode_flops = pb.flops
        self.encode_plan = pb.build()
        if arena is not None:
            # encode-only twin in the arena's buffers: every kernel of this op list has run before (the arena's own plan warmed it
            # up), nothing was allocated, so the graph is captured straight away on the lane's stream (thread-local capture mode:
            # work in flight on this and other streams is not disturbed, nothing executes here)
            assert len(pb.alloc_log) == len(arena.pb.alloc_log) and self.n_encode_ops == arena.n_encode_ops, \
                "arena replay: the two builds made different allocation / op sequences"
            if cap.use_graph:
                self.encode_plan.capture(stream or cap.stream)
            return
        # ---------------- decoder step plan (for a single micro-batch; batches of several micro-batches decode through _DecodePlans)
        self._build_step(cap, B, max_new, S, self.cross_kv, pb.ws)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)      # allocations / uploads ran on the current stream: order them before cap.stream
        if cap.use_graph:
            self.reset()
            self.encode_plan.run(cap.stream); self.step_plan.run(cap.stream)
            cap.stream.synchronize()
            self.encode_plan.capture(cap.stream)
            self.step_plan.capture(cap.stream)
            cap.stream.synchronize()

    def rows_for(self, cap: "Florence2Captioner", n: int) -> int:
        """Row count of the graph a micro-batch of n < B crops runs in this plan set's buffers.  Building a twin costs what building a
        plan costs on the host (~900 op descriptors + a graph capture: a few hundred ms), so a stream whose remainder differs from batch
        to batch must not build one per batch (measured: the mixed-resolution stream fell from 8.1 to 5.8 screenshots/s when every count
        got its own graph).  Policy: a count that is cached, or that the captioner saw for the previous remainder too (a steady load:
        the bench's 89, a service under constant traffic), runs EXACTLY; any other count runs the twin of the next capacity of the
        ladder (8, 16, 32, 64, 96 — at most five twins per plan set, the old bucket plans without their buffers)."""
        last, cap._last_remainder = getattr(cap, "_last_remainder", None), n
        if n in self._row_plans or n == last:
            return n
        return min(cap.bucket(n), self.B)

    def encode_rows(self, cap: "Florence2Captioner", n: int, stream=None) -> "_CaptionPlans":
        """encode-only plan set of exactly n <= B rows in this plan set's buffers (built and captured on first use; the
        `cap.row_graphs` most recently used row counts stay).  The caller issues it on the stream that orders this plan set's uses."""
        assert self.arena is None and 1 <= n <= self.B
        if n == self.B:
            return self
        cp = self._row_plans.pop(n, None)
        if cp is None:
            while len(self._row_plans) >= max(1, int(cap.row_graphs)):
                if dev_sync := (cap.device.type == "cuda"):
                    torch.cuda.synchronize(cap.device)                  # the evicted graph may still be executing (rare: `row_graphs` distinct counts)
                self._row_plans.pop(next(iter(self._row_plans)))
            cp = _CaptionPlans(cap, n, self.R, self.T - 1, arena=self, stream=stream)
            cap.row_graph_builds = getattr(cap, "row_graph_builds", 0) + 1
        self._row_plans[n] = cp                                         # most recently used last
        return cp


# ------------------------------------------------------------------------------------------ public objects
class _Config(SimpleNamespace):
    pass


class Florence2Captioner:
    """Duck-types the `model` half of the reference's caption_model_processor dict
    (ref:util/utils.py:108-125): `.config.name_or_path`, `.config.model_type`, `.device`, `.generate`."""
    # plan composition switches (class attributes: the tests build the round-2 composition by overriding them)
    fuse_dwln = True          # x + dwconv(x) -> LayerNorm as one strip kernel (DaViT stages 0-2)
    attn_split_out = True     # attention kernels write format B for the projection GEMM themselves
    fuse_mlp = True           # fc1 + GELU + fc2 + residual of the C = 128 stage as ONE kernel (OMNI_OP_MLP_FUSED): no hidden tensor in HBM
    exact_rows = os.environ.get("OMNI_EXACT_ROWS", "1") != "0"   # merged decode: the remainder micro-batch of a caption batch encodes exactly
                              # its own rows, as a second hipGraph over the buffers of the lane's full-capacity plan (`_CaptionPlans.encode_rows`),
                              # instead of a padded bucket plan with its own activations; `row_graphs` such graphs stay per plan set
    row_graphs = int(os.environ.get("OMNI_CAPTION_ROW_GRAPHS", "24"))
    patch_rows = os.environ.get("OMNI_PATCH_ROWS", "1") != "0"   # first patch embedding in row-patch form on the split-f16 kernel (f32 plans)
    reuse_activations = True  # scratch tensors of a DaViT stage are released at its end and back the tensors of the later stages
                              # (PlanBuilder.release): same kernels, same order, different addresses — a rank's plan sets hold 89 GB
                              # instead of 176 GB of HBM at the same speed (profiles/r4_s2_candidates_ab.txt).  With it `stage_out[:3]` are
                              # no longer valid after the encode plan: the bisection taps (tools/archive/r3_bisect.py) turn it off

    def __init__(self, model_dir, device=None, precision: Optional[str] = None, resolution: Optional[int] = None):
        device = L.require_device(device, "Florence2Captioner")
        L.lib()
        self.device = device
        precision = precision or os.environ.get("OMNI_PRECISION", "f32")
        self.dtype = L.F32 if precision == "f32" else L.F16
        # 768 = the reference's CPU-path crop resolution (parity target); 64 = its cuda branch (do_resize=False)
        self.resolution = int(resolution or os.environ.get("OMNI_CAPTION_RES", "768"))
        self.w = FlorenceWeights(model_dir)
        self.config = _Config(name_or_path=str(model_dir) if "florence" in str(model_dir).lower() else f"florence:{model_dir}",
                              model_type="florence2")
        self.use_graph = os.environ.get("OMNI_HIPGRAPH", "1") != "0"    # read when a plan is built: tools set the attribute to profile eagerly
        self.stream = self._lane_stream(0)
        self._wcache = {}
        self._plans = {}
        self.max_new_tokens = 20
        self.early_exit_every = 5        # poll the all-rows-finished flag every N decode steps (0 = always run max_new_tokens steps)
        self._lut = None
        self._lock = L.DeviceLock(self.device, reentrant=True)   # one caption batch at a time per model; makes this GPU the thread's current device

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    @staticmethod
    def bucket(n: int) -> int:
        """plan capacity for a micro-batch of n crops (rows beyond n are computed and ignored): powers of two plus 96 — the bench
        step's 349 crops are 128 + 128 + 93, and a 96-row plan for that tail computes 8 % fewer rows per step than a third 128-row
        one.  Every capacity is one more resident plan (~25 GB of activations at 128 rows, 768x768 crops — 60 GB without activation reuse; at most OMNI_CAPTION_PLAN_GB
        stay resident), which is why the ladder is not finer; OMNI_CAPTION_BUCKETS overrides it."""
        for b in _BUCKETS:
            if n <= b:
                return b
        return _BUCKETS[-1]

    @staticmethod
    def decode_bucket(n: int) -> int:
        """row capacity of the merged decode plan for n crops: multiples of 32 (the unused tail rows are computed and ignored).  Rounds
        3-5 rounded to 128: at the benched load (345 crops) that decoded 384 rows, i.e. 11 % of every per-row decode kernel —
        cross-attention streams 21.6 MB of K / V per row and step — was padding (50.4 vs 48.0 ms per batch measured in
        round 3's probe).  The micro-batches copy exactly their own n rows into the plan (`_encode_into`), so any capacity >= n works;
        32 keeps the number of distinct plans (8 GB each at 384 rows) small on a stream with varying crop counts."""
        return max(128, (n + 31) // 32 * 32)

    # ---- resident plan sets: ONE cache for encode and decode plans, bounded by BYTES (and, secondarily, by count), LRU, never
    # evicting what the batch being issued has already taken.  A 128-row plan set at 768x768 crops holds ~25 GB of activations (60 GB without activation reuse), a
    # 384-row decode plan ~8 GB: a bound counted in plans either thrashes on a stream with varying crop counts (every eviction is a
    # device-wide synchronise + a rebuild of tens of GB + a graph capture) or lets the resident set grow past the HBM.
    def begin_batch(self):
        """Called by the pipeline before it takes the plans of one caption batch: everything taken from here on is pinned until the
        next call (at most 2 lane plans + 1 remainder plan + 1 decode plan)."""
        self._epoch = getattr(self, "_epoch", 0) + 1

    def _wcache_bytes(self) -> int:
        """device bytes of the packed weights shared by all plans of this model (tensors, or tuples of tensors / None)"""
        n = 0
        for v in self.__dict__.get("_wcache", {}).values():
            for t in (v if isinstance(v, (tuple, list)) else (v,)):
                if isinstance(t, torch.Tensor):
                    n += t.numel() * t.element_size()
        return n

    def plan_cache_bytes(self) -> int:
        return sum(m[0] for k, m in self.__dict__.get("_plan_meta", {}).items() if k in self._plans)

    def clear_plans(self):
        """drop every resident plan set (the caller has synchronised the captioner's streams)"""
        self._plans.clear()
        self.__dict__.get("_plan_meta", {}).clear()

    def _cached_plan(self, key, build):
        meta = self.__dict__.setdefault("_plan_meta", {})
        epoch = getattr(self, "_epoch", 0)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)            # most recently used last
            meta.setdefault(key, [0, epoch])[1] = epoch        # (an entry placed in _plans from outside has no record: size unknown)
            return self._plans[key]
        max_bytes = int(float(os.environ.get("OMNI_CAPTION_PLAN_GB", "200")) * 2 ** 30)
        max_plans = int(os.environ.get("OMNI_MAX_CAPTION_PLANS", "16"))
        base = key[:4] if key[0] == "dec" else key[:3]         # the key without its slot: twins have the same size
        sizes = self.__dict__.setdefault("_plan_sizes", {})    # bytes of every plan set ever built (survives eviction): the estimate
        est = sizes.get(base, 0)                               # for the one about to be built
        if not est and sizes:
            # a shape never built before: scale the bytes-per-row of the known shape of the same kind and resolution (activations are
            # linear in the row count) — so the byte bound also guards FIRST builds of non-default buckets / lanes / resolutions
            same = [(b, v) for b, v in sizes.items() if (b[0] == "dec") == (base[0] == "dec") and b[-2] == base[-2] and v]
            if same:
                b0, v0 = max(same, key=lambda bv: bv[1])
                rows0, rows = (b0[1] if b0[0] == "dec" else b0[0]), (base[1] if base[0] == "dec" else base[0])
                est = int(v0 * rows / max(rows0, 1))
        def over():
            return len(self._plans) >= max_plans or (self._plans and self.plan_cache_bytes() + est > max_bytes)
        while over():
            victim = next((k for k in self._plans if meta.get(k, [0, None])[1] != epoch), None)
            if victim is None:
                break                                          # everything resident belongs to the batch being issued
            torch.cuda.synchronize(self.device)                # work of any of the captioner's streams may still use the evicted plan's buffers
            self._plans.pop(victim)
            meta.pop(victim, None)
            self.plan_evictions = getattr(self, "plan_evictions", 0) + 1
            L.gc_after_eviction()
        on_gpu = self.device.type == "cuda" and torch.cuda.is_available()
        before = torch.cuda.memory_allocated(self.device) if on_gpu else 0
        w_before = self._wcache_bytes()
        import contextlib
        with (torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()):
            obj = build()
        self._plans[key] = obj
        # the plan's OWN bytes: the first build of a model also uploads the shared weights (the weight cache outlives every plan and
        # is not freed by an eviction), which must not be booked on whichever plan happened to come first
        grown = (torch.cuda.memory_allocated(self.device) if on_gpu else 0) - before - (self._wcache_bytes() - w_before)
        meta[key] = [max(0, grown), epoch]
        sizes[base] = meta[key][0]
        return obj

    @torch.inference_mode()
    def decode_plans(self, B, R, max_new, slot=0) -> _DecodePlans:
        """slot: the pipelined stream (pipeline.py::parse_stream) decodes batch i on its own HIP stream while batch i+1 encodes, so it
        alternates between two decode plans (8 GB of cross-attention K/V each at 384 rows, 768x768 crops)."""
        key = ("dec", B, R, max_new) if slot == 0 else ("dec", B, R, max_new, slot)
        return self._cached_plan(key, lambda: _DecodePlans(self, B, R, max_new))

    @torch.inference_mode()
    def plans(self, B, R, max_new, slot=0) -> _CaptionPlans:
        """slot 1 = a second, independent set of buffers of the same capacity: the pipelined stream (pipeline.py::parse_stream) keeps two
        128-crop micro-batches in flight on two HIP streams (~25 GB of activations each at 768x768 with activation reuse, 60 GB without)."""
        key = (B, R, max_new) if slot == 0 else (B, R, max_new, slot)
        return self._cached_plan(key, lambda: _CaptionPlans(self, B, R, max_new))

    # ---- merged decode (several micro-batches): encode only, cross-KV into rows [row0, row0 + n) of the decode plan
    def _encode_into(self, cp: _CaptionPlans, n: int, dec: _DecodePlans, row0: int, stream=None):
        """encode on `stream` (default: the captioner's first stream; the caller made it current)."""
        stream = stream or self.stream
        if n < cp.B and self.exact_rows:
            cp = cp.encode_rows(self, cp.rows_for(self, n), stream)     # n rows (or the ladder capacity above n) in cp's buffers; cross_kv
                                                                        # below: the first n rows of the same tensors
        (cp.encode_plan.replay if self.use_graph else cp.encode_plan.run)(stream)
        for src, dst in zip(cp.cross_kv, dec.cross_kv):
            dst.t[row0:row0 + n].copy_(src.t[:n], non_blocking=True)

    def _decode_merged(self, dec: _DecodePlans, n: int, max_new: int, stream=None) -> torch.Tensor:
        """the 20 decode steps over all rows of `dec` on `stream` (default: the captioner's stream; the caller made it current)."""
        stream = stream or self.stream
        run = dec.step_plan.replay if self.use_graph else dec.step_plan.run
        for _ in range(max_new):
            run(stream)
        return dec.ids[:n].clone()                 # stream-ordered snapshot (read back by the caller)

    def _lane_stream(self, k):
        # Measured and not kept (round 4, profiles/r4_s2_candidates_ab.txt): CU-masked lane streams (hipExtStreamCreateWithCUMask) and a
        # two-stream replay with the GEMMs on one CU set and the HBM-bound kernels on the other — 754-1079 ms per bench step against
        # 692 on plain streams, for every partition tried.
        return torch.cuda.Stream(device=self.device)

    @property
    def stream2(self):
        """second encode lane of the pipelined stream: while one 128-crop micro-batch is in its MFMA-bound GEMMs the other one's
        HBM-bound kernels (depthwise conv + LayerNorm, attention, short-K GEMMs) fill the wave slots the GEMM blocks leave free."""
        if getattr(self, "_stream2", None) is None:
            self._stream2 = self._lane_stream(1)
        return self._stream2

    def encode_lane(self, k):
        """stream of encode lane k: 0 / 1 = `stream` / `stream2` (the default pipeline's two lanes); further lanes on demand."""
        if k == 0:
            return self.stream
        if k == 1:
            return self.stream2
        extra = self.__dict__.setdefault("_extra_lanes", {})
        if k not in extra:
            extra[k] = self._lane_stream(k)
        return extra[k]

    @property
    def dec_stream(self):
        """second HIP stream of the captioner: decode steps of batch i (launch-bound GEMMs over a few hundred rows + the HBM-bound
        cross-attention) overlap the MFMA-bound encode of batch i+1 (pipeline.py::parse_stream)."""
        if getattr(self, "_dec_stream", None) is None:
            self._dec_stream = self._lane_stream(2)
        return self._dec_stream

    # ---- decode loop shared by both entry points
    def _run(self, cp: _CaptionPlans, n: int, max_new: int, defer: bool = False) -> torch.Tensor:
        run = (lambda p: p.replay(self.stream)) if self.use_graph else (lambda p: p.run(self.stream))
        run(cp.encode_plan)
        poll = 0 if defer else self.early_exit_every
        for t in range(max_new):
            run(cp.step_plan)
            # hf stops as soon as every row has emitted EOS (generation/utils.py:2936): poll the device flags every few steps
            # (a 4-byte read-back) instead of always paying max_new steps; the deferred (batched-stream) path never syncs
            if poll and (t + 1) % poll == 0 and t + 1 < max_new and bool(cp.finished[:n].min().item()):
                break
        if defer:                          # stream-ordered snapshot; the caller reads it back later (no sync here)
            return cp.ids[:n].clone()
        return self._finish_ids(cp.ids[:n].cpu().long())     # synchronises the stream

    def _finish_ids(self, ids: torch.Tensor) -> torch.Tensor:
        n = ids.shape[0]
        # hf stops as soon as every row has emitted EOS (generation/utils.py:2936): trim the all-pad tail
        T = ids.shape[1]
        done_at = T
        seen = torch.zeros(n, dtype=torch.bool)
        for t in range(1, T):
            seen |= ids[:, t] == self.w.eos
            if bool(seen.all()):
                done_at = t + 1
                break
        return ids[:, :done_at]

    @torch.inference_mode()
    def generate(self, input_ids=None, pixel_values=None, max_new_tokens=20, num_beams=1, do_sample=False, **kw):
        """hf-compatible entry point (ref:util/utils.py:125).  pixel_values: [B,3,R,R] float (NCHW)."""
        if num_beams != 1 or do_sample:
            raise NotImplementedError("greedy decoding only (the reference calls num_beams=1, do_sample=False)")
        with self._lock:
            return self._generate_locked(pixel_values, max_new_tokens)

    def _generate_locked(self, pixel_values, max_new_tokens):
        Bn, _, R, R2 = pixel_values.shape
        assert R == R2
        out = []
        for s in range(0, Bn, 128):
            chunk = pixel_values[s:s + 128]
            n = chunk.shape[0]
            cp = self.plans(self.bucket(n), R, max_new_tokens)
            if chunk.is_cuda:                                   # pixel_values the caller is still producing on its own stream
                self.stream.wait_stream(torch.cuda.current_stream(chunk.device))
            with torch.cuda.stream(self.stream):
                cp.reset()
                cp.x_in.t[:n, :, :, :3] = chunk.to(self.device).permute(0, 2, 3, 1).to(cp.x_in.t.dtype)
                out.append(self._run(cp, n, max_new_tokens))
        T = max(o.shape[1] for o in out)
        res = torch.full((Bn, T), self.w.pad, dtype=torch.long)
        o0 = 0
        for o in out:
            res[o0:o0 + o.shape[0], :o.shape[1]] = o
            o0 += o.shape[0]
        return res

    @torch.inference_mode()
    def caption_crops(self, image_u8: torch.Tensor, boxes_px: List[List[int]], max_new_tokens=20, batch_size=128):
        """Fused fast path: crops are cut, resized (cv2-bilinear 64x64, then Pillow-bicubic to R on the
        768 path) and normalised on device from the HBM-resident screenshot (ref:util/utils.py:97-123)."""
        with self._lock:
            return self._caption_crops_locked(image_u8, boxes_px, max_new_tokens, batch_size)

    def _caption_crops_locked(self, image_u8, boxes_px, max_new_tokens, batch_size):
        n_all = len(boxes_px)
        R = self.resolution
        outs = []
        H, W = image_u8.shape[:2]
        if self._lut is None:
            self._lut = torch.from_numpy((np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32)).to(self.device)
            if R != 64:
                b, k = L.resample_coeffs(64, R, 1)
                self._bic = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
        batch_size = max(1, min(int(batch_size), 128))      # plan capacity: buckets stop at 128 crops (the reference's default batch)
        if image_u8.is_cuda:                                 # a screenshot the caller is still uploading / drawing on its own stream
            self.stream.wait_stream(torch.cuda.current_stream(image_u8.device))
        for s in range(0, n_all, batch_size):
            boxes = boxes_px[s:s + batch_size]
            n = len(boxes)
            cp = self.plans(self.bucket(n), R, max_new_tokens)
            with torch.cuda.stream(self.stream):
                cp.reset()
                bx = torch.tensor(boxes, dtype=torch.int32).to(self.device, non_blocking=True)
                c64 = torch.empty((n, 64, 64, 3), dtype=torch.uint8, device=self.device)
                tmp = torch.empty((n, 64, R, 3), dtype=torch.uint8, device=self.device) if R != 64 else None
                bb, kk, ks = self._bic if R != 64 else (None, None, 0)
                op = L.make_op(L.OP_CROP_RESIZE, self.dtype,
                               p=[image_u8.data_ptr(), bx.data_ptr(), c64.data_ptr(), tmp.data_ptr() if tmp is not None else None,
                                  cp.x_in.ptr, bb.data_ptr() if bb is not None else None, kk.data_ptr() if kk is not None else None,
                                  self._lut.data_ptr()],
                               i={0: n, 1: H, 2: W, 3: R, 4: ks, 13: cp.x_in.ld},
                               f={0: CLIP_MEAN[0], 1: CLIP_MEAN[1], 2: CLIP_MEAN[2], 3: CLIP_STD[0], 4: CLIP_STD[1], 5: CLIP_STD[2]})
                L.launch(op, self.stream)
                outs.append(self._run(cp, n, max_new_tokens))
        if not outs:
            return torch.zeros((0, 1), dtype=torch.long)
        T = max(o.shape[1] for o in outs)
        res = torch.full((n_all, T), self.w.pad, dtype=torch.long)
        o0 = 0
        for o in outs:
            res[o0:o0 + o.shape[0], :o.shape[1]] = o
            o0 += o.shape[0]
        return res
