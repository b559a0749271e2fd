"""-m gpu: parity of the step bench.py TIMES — `DLKABlockStack(2, SYNAPSE_STAGES)`, 21 blocks at BASELINE.json config 3's sizes, weight gradients on the side stream,
two alternating workspaces, library-internal fork streams, replayed from a hipGraph (bench.py main(): the eager warm-up, the side-stream warm-up, the capture) — against
  (1) the per-block entry points (`dlka_lka3d_attention_tokens_forward/backward`, one stream, private scratch) on the SAME x / grad_y, for the first and the last block
      of every chain: y, the predicted offsets, grad_x and all 14 parameter gradients;
  (2) the CPU oracle (oracle/blocks.py) with the flips-counted + same-cells protocol of tests/parity.py, for a 32^3 block at each end of the step and a 4^3 block.
Every other test of the engine runs toy volumes; a workspace race that needs 32^3-sized kernels beside each other only shows here (missing #1).
Reference: model_components.py:33-39,127-131 (the chains), transformerblock.py:664-673 (the block)."""
import pytest
import torch

from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"



@pytest.fixture(scope="module", autouse=True)
def hip_backend(oracle):
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert torch.cuda.is_available()
    _lib.get_lib()
    yield


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_benchmarked_step_full_size_vs_per_block_entry_and_oracle(dtype):
    from deformablelka_amd.stack import DLKABlockStack, SYNAPSE_STAGES
    B = 2
    st = DLKABlockStack(B, SYNAPSE_STAGES, device=DEV, dtype=dtype, seed=1234, data_seed=4321)   # bench.py: the same seeds
    assert st._overlap, "the benchmarked configuration runs its weight gradients on the side stream"
    st.forward_backward()          # eager warm-up: records and seals the fold plan
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        st.forward_backward()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(graph):
        st.forward_backward()
    assert st._fin_sealed
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert st.health()["finite"]

    picked, i = [], 0
    for chain in st.chains:
        picked += sorted({i, i + len(chain) - 1})
        i += len(chain)
    assert len(picked) == 14 and picked[0] == 0 and picked[-1] == len(st.blocks) - 1
    # the oracle, on three blocks of the replayed step: the encoder's first 32^3 block (first of the forward pass), the decoder's last 32^3 block (first of the
    # backward pass: its weight gradients are the first on the side stream), and the last block of the 4^3 chain (chain-interior x, tap-split kernels)
    stage3 = next(i_ for i_, b in enumerate(st.blocks) if b.C == 256) + 2
    # (bf16 storage: the per-block entry points only — the oracle comparison of the bf16 block is test_lka3d_tokens_bf16_headline_shapes_vs_oracle's)
    parity.check_stack_step(st, picked, (0, len(st.blocks) - 1, stage3) if dtype == torch.float32 else ())
