"""not-gpu: the CLI mirror (flags and defaults of the reference's main.py:19-77), point-cloud I/O
round trips and the checkpoint format."""
import os

import numpy as np
import torch

from conftest import pkg, sphere


def test_cli_flags_and_defaults_match_reference():
    main = pkg("main")
    p = main.build_parser()
    d = vars(p.parse_args([]))
    expect = dict(phase='test', gpu=0, id='demo', log_dir='./model', model='model_microscope', root_dir='../',
                  result_dir=None, ckpt=None, num_point=None, num_shape_point=None, up_ratio=16, max_epoch=160,
                  batch_size=16, h5_data=None, record_data=None, test_data=None, lr_init=0.0005,
                  restore_epoch=None, stage_steps=15000, step_ratio=2, patch_num_ratio=3, jitter=False,
                  jitter_sigma=0.0025, jitter_max=0.005, drop_out=1.0, knn=32, dense_n=3, block_n=3, fm_knn=5,
                  growth_rate=12, cd_threshold=2.0, fidelity_weight=50.0)
    assert d == expect
    assert main.get_stage_progress(0, 15000) == (0, 0.5)
    assert main.get_stage_progress(15000, 15000) == (1, 0.0)
    f = p.parse_args(["--num_point", "312", "--num_shape_point", "5000", "--jitter", "--drop_out", "0.5"])
    assert main.result_path_of(f, 312, 5000, "./model/demo").endswith("x16/p312_s5000_s00025_d050")


def test_ply_and_xyz_roundtrip(tmp_path):
    pcu = pkg("utils.pc_utils")
    pts = sphere(0, 1000)[0]
    ply = os.path.join(str(tmp_path), "sub", "a.ply")
    pcu.save_ply(pts, ply)
    np.testing.assert_array_equal(pcu.load(ply), pts)
    raw = open(ply, "rb").read()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000\nproperty float x\n")
    xyz = os.path.join(str(tmp_path), "a.xyz")
    np.savetxt(xyz, pts)
    np.testing.assert_allclose(pcu.load(xyz), pts, rtol=1e-6)
    np.random.seed(0)
    padded = pcu.load(xyz, 1200)                      # pad by random duplication (reference :229-235)
    assert padded.shape == (1200, 3) and (padded[:1000] == pcu.load(xyz)).all()
    assert all((padded[i] == padded[:1000]).all(1).any() for i in range(1000, 1200))
    assert pcu.load(xyz, 300).shape == (300, 3)
    n, c, r = pcu.normalize_point_cloud(pts[None] * 3 + 1)
    assert abs(np.linalg.norm(n[0], axis=1).max() - 1) < 1e-6 and c.shape == (1, 1, 3) and r.shape == (1, 1, 1)


def test_checkpoint_format_roundtrip(tmp_path):
    pu, ups = pkg("utils.pytorch_utils"), pkg("network.upsampler")
    torch.manual_seed(1)
    net = ups.Net(max_up_ratio=4, step_ratio=2, knn=32)
    path = pu.save_network(net, str(tmp_path), "model", epoch_label="20", step="1234")
    assert path.endswith("model_20.pth")
    blob = torch.load(path)
    assert set(blob.keys()) == {"states", "step"} and blob["step"] == "1234"
    blob["states"]["not.a.parameter"] = torch.zeros(1)          # extra keys are dropped on load
    torch.save(blob, path)
    net2 = ups.Net(max_up_ratio=4, step_ratio=2, knn=32)
    assert pu.load_network(net2, path) == 1234
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)


def test_checkpoint_as_pickled_numpy_dict(tmp_path):
    """The reference's load_network also takes a pickled numpy dict (utils/pytorch_utils.py:24-27: any path not
    ending in 'pth' goes through np.load(...).item()): same layout, arrays or tensors under 'states'."""
    pu, ups = pkg("utils.pytorch_utils"), pkg("network.upsampler")
    torch.manual_seed(2)
    net = ups.Net(max_up_ratio=2, step_ratio=2, knn=32)
    record = {"states": {k: v.numpy() for k, v in net.state_dict().items()}, "step": 77}
    record["states"]["stale.entry"] = np.zeros(3, np.float32)
    path = str(tmp_path / "model.npy")
    np.save(path, np.array(record, dtype=object), allow_pickle=True)
    net2 = ups.Net(max_up_ratio=2, step_ratio=2, knn=32)
    assert pu.load_network(net2, path) == 77
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    # a file without a step counts from 0
    del record["step"]
    np.save(path, np.array(record, dtype=object), allow_pickle=True)
    assert pu.load_network(net2, path) == 0
