"""Sample socket power / engine clock (rocm-smi) while the split decoder runs back to back; prints the
idle figures, the loaded figures and the kernel's rate.  Evidence for DESIGN 3b.1 (is the kernel power-limited?)."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True).stdout
    return out.strip()


def main():
    print("idle: " + smi().strip().split("\n")[-1], flush=True)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.time(), smi()))
            time.sleep(0.2)

    import numpy as np
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8,
                   num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    if os.environ.get("ZS_NO_GUARD"):          # ablation builds produce garbage: keep the fp32 re-evaluation out
        net.envelope_guard = False
    latent = torch.from_numpy(syn.seeded_latent(0, 1)).cuda()
    axis = torch.linspace(-0.6, 0.6, 129, device="cuda")
    st = net.prepare(latent)
    run = lambda: net.query_grid(latent, axis, state=st)          # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6.0:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        n += 10
    dt = time.time() - t0
    stop.set()
    th.join()
    pw, ck = [], []
    for t, s in samples[3:]:
        row = s.strip().split("\n")[-1].split(",")
        pw.append(float(row[-1]))
        ck.append(float(re.sub(r"[^0-9.]", "", row[7])))
    print("loaded: %s  %d launches in %.2f s = %.3f ms each; socket power %.0f W (max %.0f), sclk %.0f MHz, %.2f J per launch"
          % (os.environ.get("ZS_LIB_PATH", "default"), n, dt, 1e3 * dt / n, sum(pw) / len(pw), max(pw), sum(ck) / len(ck),
             sum(pw) / len(pw) * dt / n))


if __name__ == "__main__":
    main()
