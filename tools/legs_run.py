import json, sys, torch
sys.path.insert(0, "/root/repo")
from tools import bench_legs as legs
dev = torch.device("cuda:0")
which = sys.argv[1:]
out = {}
if "att" in which:
    out["encoder_att"] = legs.encoder_att_leg(dev)
if "enc" in which:
    out["encoder"] = legs.encoder_leg(dev, cpu=False)
if "vox256" in which or "eval" in which:
    import numpy as np
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    if "vox256" in which:
        out["vox256"] = legs.vox256_leg(dev, net)
    if "eval" in which:
        out["chamfer_l1"] = legs.eval_leg(dev, net, sd)
print(json.dumps(out, indent=1))
