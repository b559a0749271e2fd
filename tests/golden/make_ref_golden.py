"""Generates tests/golden/d3d_reference_vectors.pt: inputs and outputs of the REFERENCE'S OWN native op
(oracle/_ref/D3D.so = 3D/dcn/src compiled unmodified by oracle/ref.mk) on the small cases of tests/ref_cases.py.

Has to run on a GPU (the reference's CPU branch is AT_ERROR, 3D/dcn/src/deform_conv.h:46,90):
    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/d3d_reference_vectors.pt'
then copy the file (and its `_2d` sibling: the 2-D cases, computed by the same op on the D = 1 embedding) to tests/golden/.  The CPU suite (tests/test_oracle_vs_reference_vectors.py) checks the C oracle against it,
which is what pins the oracle to the reference's arithmetic where no GPU exists."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import ref_cases  # noqa: E402


def main(out_path):
    assert torch.cuda.is_available()
    blob = {"_meta": {"made_by": "tests/golden/make_ref_golden.py", "source": "oracle/_ref/D3D.so (reference 3D/dcn/src, hipcc gfx950)",
                      "torch": str(torch.__version__), "device": torch.cuda.get_device_name(0)}}
    for name, case in ref_cases.SMALL.items():
        t = ref_cases.make(case)
        out, gi, goff, gw, gb = ref_cases.run_ref(t, "cuda:0")
        blob[name] = {"case": case, "x": t["x"], "off": t["off"], "w": t["w"], "b": t["b"], "go": t["go"],
                      "out": out, "grad_input": gi, "grad_offset": goff, "grad_weight": gw, "grad_bias": gb}
        print(name, tuple(out.shape), float(out.abs().max()))
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    torch.save(blob, out_path)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")
    # 2-D: the same op on the D = 1 embedding (tests/ref_cases.py) = reference-arithmetic vectors for the torchvision-semantics operator
    blob2 = {"_meta": dict(blob["_meta"], embedding="D = 1, kd = 1, pad_d = 0, zero depth offsets, zero bias (tests/ref_cases.py: embed2d)")}
    for name in ref_cases.SMALL_2D:
        case = ref_cases.CASES_2D[name]
        t = ref_cases.make2d(case)
        (out, gi, goff, gw), gd = ref_cases.run_ref2d(t, "cuda:0")
        blob2[name] = {"case": case, "x": t["x"], "off": t["off"], "w": t["w"], "go": t["go"], "out": out, "grad_input": gi, "grad_offset": goff,
                       "grad_weight": gw}
        print("2d", name, tuple(out.shape), float(out.abs().max()))
    out2 = out_path.replace(".pt", "_2d.pt")
    torch.save(blob2, out2)
    print("wrote", out2, os.path.getsize(out2), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "d3d_reference_vectors.pt"))
