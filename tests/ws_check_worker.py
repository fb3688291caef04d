"""Child process of tests/test_gpu_widened.py::test_workspace_contract_is_checked_in_the_experiments_build: the experiments build with
NAFAE_WS_CHECK=1 must refuse (NAFAE_EINVAL) a stream-K conv whose workspace counters are not zero -- the contract of
include/nafae_hip.h since round 4 ("first 64 KB zero when the first call on a workspace starts") -- and accept the same call once they
are, in exact fp32 and on the bf16 engine.  Prints OK."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from nafae_amd import _lib, ops
    assert _lib.LIB_PATH.endswith("_exp.so") and os.environ.get("NAFAE_WS_CHECK") == "1"
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(64, 14, 14, 512, device="cuda", generator=g)             # 1.53 tiles per CU: the stream-K schedule
    w = torch.randn(512, 3, 3, 512, device="cuda", generator=g) * 0.02
    b = torch.zeros(512, device="cuda")
    ref = ops.conv3x3_relu(x, w, b)                                          # zeroed-at-allocation workspace: accepted
    for t in ops._conv_ws.values():
        t[:8] = 1                                                            # a caller on the old contract / after an aborted launch
    try:
        ops.conv3x3_relu(x, w, b)
        raise SystemExit("non-zero counters were not refused (fp32)")
    except ops.NafaeOpError as e:
        assert "-1" in str(e), e
    x28 = torch.randn(64, 28, 28, 512, device="cuda", generator=g)          # bf16 engine: the 28^2 layers take its stream-K schedule
    xp, wp = ops.split_bf16(x28, True, True), ops.split_bf16(w, True, True)
    ops.conv3x3_bf16(xp, wp, b, relu=True)                                   # (allocates / grows the shared workspace, zeroed)
    for t in ops._conv_ws.values():
        t[:8] = 1
    try:
        ops.conv3x3_bf16(xp, wp, b, relu=True)
        raise SystemExit("non-zero counters were not refused (bf16)")
    except ops.NafaeOpError as e:
        assert "-1" in str(e), e
    for t in ops._conv_ws.values():
        t[:8] = 0
    assert torch.equal(ops.conv3x3_relu(x, w, b), ref)
    ops.conv3x3_bf16(xp, wp, b, relu=True)
    torch.cuda.synchronize()
    print("OK")


if __name__ == "__main__":
    main()
