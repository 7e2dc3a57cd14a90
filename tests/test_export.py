"""CPU: the robot-side export (robocupvision_amd/export.py) -- flat float64 weight dump (paramSave.py) against the reference's
own dump of the same twice-stepped PB_FCN (hash in tests/golden/pbfcn.json; the state is rebuilt here with the CPU oracle), and the
net.cfg layer list against the reference's weights/net.cfg when that file is present (it is data, not copied into this repo)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import cpu_reference as O
import robocupvision_amd.model as M
from robocupvision_amd.export import flat_params, net_cfg, saveParams


def test_flat_params_match_reference_dump(pb_meta, tmp_path):
    m = pb_meta["pbfcn_s_2x48x64"]
    old = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        torch.manual_seed(12345678)
        model = M.PB_FCN(32, 5, 1, False, 0)
        st = O.PBTrainState(model.state_dict(), False)
        x, t = O.synthetic_batch(m["B"], m["H"], m["W"])
        O.pb_train_step(st, x, t)
        O.pb_train_step(st, x, t)
    finally:
        torch.set_num_threads(old)
    model.load_state_dict({k: v.detach() for k, v in st.sd.items()})
    g = pb_meta["saveParams"]
    flat = flat_params(model)
    assert flat.dtype == np.float64 and flat.size == g["count"]
    assert hashlib.sha256(flat.tobytes()).hexdigest() == g["sha256"]
    assert [float(v) for v in flat[:8]] == g["head"] and [float(v) for v in flat[-8:]] == g["tail"]
    skip = flat_params(model, skipClassifier=True)
    assert skip.size == g["count_skip_classifier"] and hashlib.sha256(skip.tobytes()).hexdigest() == g["sha256_skip_classifier"]
    saveParams(str(tmp_path / "w"), model, "weights.dat")
    assert np.array_equal(np.fromfile(str(tmp_path / "w" / "weights.dat")), flat)


def test_net_cfg_layer_list():
    torch.manual_seed(0)
    text = net_cfg(M.PB_FCN(32, 5, 1, False, 0))
    lines = [l.strip() for l in text.strip().splitlines() if l.strip()]
    assert lines[0] == "[net]" and lines[-1] == "[softmax]"
    assert [l for l in lines if l.startswith("from=")] == ["from=6", "from=3", "from=1"]       # closing layers of conv2, conv1, conv0
    assert sum(l == "[convolutional]" for l in lines) == 12 and sum(l == "[transposedconv]" for l in lines) == 3
    ref_path = "/root/reference/weights/net.cfg"
    if not os.path.exists(ref_path):
        pytest.skip("reference checkout not present on this machine")
    with open(ref_path) as f:
        ref = [l.strip() for l in f.read().strip().splitlines() if l.strip()]
    assert lines == ref
    big = net_cfg(M.PB_FCN(32, 5, 1, True, 0))
    assert big.count("[transposedconv]") == 4 and "height=240" in big
