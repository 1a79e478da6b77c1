"""Golden vectors for the tiled VAE from the REFERENCE's own VAEHook (authoring container only)."""
import copy
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
import numpy as np
import torch
import yaml

import s2_common as S
from oracle import seeded, tilevae_oracle as TO

torch.set_num_threads(8)


@torch.no_grad()
def main():
    import utils.devices as devices
    devices.device = torch.device("cpu")
    devices.get_optimal_device = lambda: torch.device("cpu")
    import utils.tilevae as TV
    from sgm.models.autoencoder import AutoencoderKLInferenceWrapper
    cfg = yaml.safe_load(open("/root/reference/model_configs/juggernautXL.yaml"))["model"]["params"]["first_stage_config"]["params"]
    cfg["ddconfig"]["attn_type"] = "vanilla-xformers"  # tilevae.py:336 needs net.attention_op
    cfg["lossconfig"] = {"target": "torch.nn.Identity"}
    fs = AutoencoderKLInferenceWrapper(**cfg).eval()
    seeded.seed_module(fs, S.WEIGHT_SEED + 1)
    sd = {"first_stage_model." + k: v.detach().clone() for k, v in fs.state_dict().items()}

    # ---- integer tile geometry
    geo = {}
    for (h, w, tile, dec) in [(256, 256, 128, False), (4096, 4096, 512, False), (1024, 3072, 512, False), (512, 512, 64, True),
                              (128, 384, 64, True), (100, 37, 24, True), (320, 200, 96, False)]:
        hook = TV.VAEHook(fs.decoder if dec else fs.encoder, tile, is_decoder=dec, fast_decoder=False, fast_encoder=False, color_fix=False)
        ins, outs = hook.split_tiles(h, w)
        geo[f"{h}x{w}_t{tile}_{'dec' if dec else 'enc'}"] = {"in": ins, "out": outs}
        assert (ins, outs) == TO.split_tiles(h, w, tile, 11 if dec else 32, dec)
    json.dump(geo, open(os.path.join(HERE, "tilevae_geometry.json"), "w"))

    # ---- GroupNormParam.summary on synthetic stats
    g = {}
    gp = TV.GroupNormParam()
    tiles = [S.rnd((2, 64, 12, 10), 1), S.rnd((2, 64, 12, 7), 2, 2.0) + 0.5, S.rnd((2, 64, 5, 10), 3, 0.3)]
    class L:  # layer with affine
        weight, bias = S.rnd((64,), 4) * 0.1 + 1, S.rnd((64,), 5) * 0.1
    for t in tiles:
        gp.add_tile(t, L)
    fn = gp.summary()
    g["summary.out1"] = fn(tiles[1].clone()).numpy()
    lsd = {"n.weight": L.weight, "n.bias": L.bias}
    print("cross_tile_norm oracle vs reference", float((torch.tensor(g["summary.out1"]) - TO.cross_tile_norm(lsd, "n", tiles, False)[1]).abs().max()))

    # ---- full hooks: encoder 256x192 image, tile 96 (pad 32) ; decoder 40x28 latent, tile 12 (pad 11)
    img = seeded.synthetic_image((1, 3, 256, 192), seed=90, smooth=3)
    enc_hook = TV.VAEHook(fs.encoder, 96, is_decoder=False, fast_decoder=False, fast_encoder=False, color_fix=False)
    fs.encoder.original_forward = fs.encoder.forward
    h = enc_hook(img)
    g["enc.out"] = h.numpy()
    ho = TO.tiled_forward(sd, img, 96, False, "first_stage_model.encoder.")
    print("tiled encoder oracle vs reference", float((h - ho).abs().max()), "range", float(h.abs().max()),
          "| tiled vs untiled", float((h - fs.encoder.original_forward(img)).abs().max()))
    zl = S.rnd((1, 4, 40, 28), 91)
    dec_hook = TV.VAEHook(fs.decoder, 12, is_decoder=True, fast_decoder=False, fast_encoder=False, color_fix=False)
    fs.decoder.original_forward = fs.decoder.forward
    zin = fs.post_quant_conv(zl)
    d = dec_hook(zin)
    g["dec.out"] = d.numpy()
    do = TO.tiled_forward(sd, zin, 12, True, "first_stage_model.decoder.")
    print("tiled decoder oracle vs reference", float((d - do).abs().max()), "range", float(d.abs().max()),
          "| tiled vs untiled", float((d - fs.decoder.original_forward(zin)).abs().max()))
    np.savez_compressed(os.path.join(HERE, "tilevae_golden.npz"), **g)
    print("done")


if __name__ == "__main__":
    main()
