"""Golden vectors for torchrl/objectives/value/utils.py (generated in the build container from the reference
imported with tools/ref_import.py).  Writes tests/golden/value_utils.npz + value_utils_manifest.json.
    python tools/gen_golden_utils.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    F = ref_import.load_functional()
    R = sys.modules[F._custom_conv1d.__module__]
    g = torch.Generator().manual_seed(77)
    arrays, cases = {}, []

    def add(fn, meta, **tensors):
        cid = f"u{len(cases):03d}"
        for k, v in tensors.items():
            arrays[f"{cid}.{k}"] = v.numpy()
        cases.append(dict(meta, id=cid, fn=fn))

    for shape, dim in (((3, 4), 0), ((3, 4), 1), ((2, 5, 7), -2), ((2, 5, 7), -1)):
        x = torch.randn(shape, generator=g)
        n = shape[-1] if dim in (0, -2) else shape[-2]
        sh = torch.randint(-9, 9, (n,), generator=g)
        add("roll_by_gather", dict(dim=dim), mat=x, shifts=sh, out=R.roll_by_gather(x, dim, sh))
    for roll in (True, False):
        for shape in ((3, 6, 2), (2, 2, 5, 1), (1, 1, 1)):
            ga = torch.rand(shape, generator=g)
            add("_make_gammas_tensor", dict(T=shape[-2], rolling_gamma=roll), gamma=ga, out=R._make_gammas_tensor(ga, shape[-2], roll))
    for B, T, Fd, per_step in ((4, 9, 5, False), (4, 9, 9, False), (3, 7, 1, False), (4, 9, 6, True), (4, 9, 9, True), (2, 1, 1, True)):
        t = torch.randn(B, 1, T, generator=g)
        f = torch.rand(B, T, Fd, 1, generator=g) if per_step else torch.rand(Fd, 1, generator=g)
        add("_custom_conv1d", dict(per_step=per_step), tensor=t, filter=f, out=R._custom_conv1d(t, f))
    for B in ((1,), (3,), (2, 2), (1, 2, 8)):
        for T in (1, 5, 40):
            done = torch.rand(*B, T, generator=g) < 0.2
            splits = R._get_num_per_traj(done)
            traj = torch.rand(*B, T, generator=g)
            padded, mask = R._split_and_pad_sequence(traj, splits, return_mask=True)
            feat = torch.rand(*B, T, 3, generator=g)
            init = torch.rand(*B, T, generator=g) < 0.3
            add("trajectories", dict(B=list(B), T=T), done=done, splits=splits, traj=traj, padded=padded, mask=mask,
                unpadded=R._inv_pad_sequence(padded, splits), feat=feat, padded_feat=R._split_and_pad_sequence(feat, splits, time_dim=-2),
                is_init=init, splits_init=R._get_num_per_traj_init(init))
    np.savez_compressed(os.path.join(OUT, "value_utils.npz"), **arrays)
    with open(os.path.join(OUT, "value_utils_manifest.json"), "w") as f:
        json.dump(dict(generator="tools/gen_golden_utils.py", reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/objectives/value/utils.py",
                       torch=torch.__version__, cases=cases), f, indent=1)
    print(len(cases), "cases;", os.path.getsize(os.path.join(OUT, "value_utils.npz")), "bytes")


if __name__ == "__main__":
    main()
