"""Captioner parity harness shared by CPU (plan interpreter) and GPU (HIP kernels) tests."""
from types import SimpleNamespace

import numpy as np
import torch

from omniparser_amd import _lib as L
from omniparser_amd import florence as FL


def hf_reference(model, pixel_values, max_new_tokens=20):
    """transformers-native Florence-2 on CPU: image features, encoder output and greedy ids."""
    cfg = model.config
    B = pixel_values.shape[0]
    with torch.inference_mode():
        feats = model.model.get_image_features(pixel_values).pooler_output
        n_img = feats.shape[1]
        ids = torch.tensor([[cfg.image_token_id] * n_img + FL.PROMPT_IDS] * B)
        emb = model.get_input_embeddings()(ids)
        mask = (ids == cfg.image_token_id).unsqueeze(-1)
        emb = emb.masked_scatter(mask, feats)
        enc = model.model.language_model.encoder(inputs_embeds=emb).last_hidden_state
        out = model.generate(input_ids=ids, pixel_values=pixel_values, max_new_tokens=max_new_tokens, num_beams=1, do_sample=False)
    return feats, enc, out


def build_cpu_plans(weights_dir, B, R, dtype=L.F32, max_new=20):
    """_CaptionPlans on CPU tensors (no kernels run) for the interpreter."""
    # every plan-composition switch of the class (its bool class attributes), so a new switch cannot be forgotten here
    switches = {k: v for k, v in vars(FL.Florence2Captioner).items() if isinstance(v, bool)}
    cap = SimpleNamespace(w=FL.FlorenceWeights(weights_dir), device=torch.device("cpu"), dtype=dtype, _wcache={},
                          use_graph=False, stream=None, **switches)
    cp = FL._CaptionPlans(cap, B, R, max_new)
    return cap, cp


def all_tensors(cap, cp):
    keep = list(cp.pb.keep) + list(cp.pd.keep)
    for v in cap._wcache.values():
        keep += [t for t in (v if isinstance(v, tuple) else (v,)) if isinstance(t, torch.Tensor)]
    return keep


def run_interp(cap, cp, pixel_values, max_new=20):
    from plan_interp import Mem, run_op
    n = pixel_values.shape[0]
    cp.reset()
    cp.x_in.t[:n, :, :, :3] = pixel_values.permute(0, 2, 3, 1).to(cp.x_in.t.dtype)
    mem = Mem(all_tensors(cap, cp))
    for op in cp.encode_plan.ops:
        run_op(op, mem)
    feats = cp.img_feat.t[:n, :, 0, :].float().clone()
    enc = cp.enc_out.t[:n, :, 0, :].float().clone()
    for _ in range(max_new):
        for op in cp.step_plan.ops:
            run_op(op, mem)
    return feats, enc, cp.ids[:n].long().clone()
