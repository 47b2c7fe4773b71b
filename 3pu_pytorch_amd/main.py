"""Command line of the reference (main.py:19-77: same flags, same defaults) over the MI355X path.

    python -m torch... no launcher needed:   python 3pu_pytorch_amd/main.py --phase test --ckpt model.pth \\
        --num_point 312 --num_shape_point 5000 --up_ratio 16 --test_data "data/*.xyz"

`--phase test` follows main.py:333-389 (load, normalise, optional jitter, patch-wise upsampling,
final FPS, de-normalise, write `<name>_input.ply` / `<name>.ply`) with the per-patch Python loop of
pc_prediction (:214-246) replaced by the batched pipeline.  `--phase train` runs Model.optimize
(model.py:53-66) with the reference's curriculum bookkeeping (main.py:118-124,141-182) on patch
pairs drawn from the data set (`--h5_data <file>`: data.py's H5Dataset, device-resident; .npz, or .hdf5
with h5py) or, with `--h5_data synthetic`, from generated sphere pairs of the same array shapes.  `--phase vis` (interactive matplotlib) is
out of scope.  Unlike the reference nothing is parsed or built at import time.
"""
import argparse
import importlib
import os
import sys
import time
from glob import glob

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))
_pkg = importlib.import_module(os.path.basename(_HERE))
Net = importlib.import_module(_pkg.__name__ + ".network.upsampler").Net
operations = importlib.import_module(_pkg.__name__ + ".network.operations")
pipeline = importlib.import_module(_pkg.__name__ + ".pipeline")
Model = importlib.import_module(_pkg.__name__ + ".model").Model
pc_utils = importlib.import_module(_pkg.__name__ + ".utils.pc_utils")
pytorch_utils = importlib.import_module(_pkg.__name__ + ".utils.pytorch_utils")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--phase', default='test', help='train or test [default: train]')
    parser.add_argument('--gpu', type=int, default=0, help='GPU to use [default: GPU 0]')
    parser.add_argument('--id', default='demo', help="experiment name, prepended to log_dir")
    parser.add_argument('--log_dir', default='./model', help='Log dir [default: log]')
    parser.add_argument('--model', default='model_microscope', help='model name')
    parser.add_argument('--root_dir', default='../', help='project root, data and h5_data diretories')
    parser.add_argument('--result_dir', help='result directory')
    parser.add_argument('--ckpt', help='model to restore from')
    parser.add_argument('--num_point', type=int, help='Point Number [1024/2048] [default: 1024]')
    parser.add_argument('--num_shape_point', type=int, help="Number of points per shape")
    parser.add_argument('--up_ratio', type=int, default=16, help='Upsampling Ratio [default: 2]')
    parser.add_argument('--max_epoch', type=int, default=160, help='Epoch to run [default: 500]')
    parser.add_argument('--batch_size', type=int, default=16, help='Batch Size during training')
    parser.add_argument('--h5_data', help='h5 file for training')
    parser.add_argument('--record_data', help='record file for training')
    parser.add_argument('--test_data', help='test data path')
    parser.add_argument('--lr_init', type=float, default=0.0005)
    parser.add_argument('--restore_epoch', type=int)
    parser.add_argument('--stage_steps', type=int, default=15000, help="number of updates per curriculums stage")
    parser.add_argument('--step_ratio', type=int, default=2, help="upscale ratio per step")
    parser.add_argument('--patch_num_ratio', type=float, default=3)
    parser.add_argument('--jitter', action="store_true", help="jitter augmentation")
    parser.add_argument('--jitter_sigma', type=float, default=0.0025, help="jitter augmentation")
    parser.add_argument('--jitter_max', type=float, default=0.005, help="jitter augmentation")
    parser.add_argument('--drop_out', type=float, default=1.0, help="drop_out ratio. default 1.0 (no drop out) ")
    parser.add_argument('--knn', type=int, default=32, help="neighbood size for edge conv")
    parser.add_argument('--dense_n', type=int, default=3, help="number of dense layers")
    parser.add_argument('--block_n', type=int, default=3, help="number of dense blocks")
    parser.add_argument('--fm_knn', type=int, default=5, help="number of neighboring points for feature matching")
    parser.add_argument('--growth_rate', type=int, default=12, help='dense block growth rate')
    parser.add_argument('--cd_threshold', default=2.0, type=float, help="threshold for cd")
    parser.add_argument('--fidelity_weight', default=50.0, type=float, help="chamfer loss weight")
    return parser


def get_stage_progress(step, stage_steps):
    """return the stage (an integer from 0) and progress (float 0~1)   (main.py:118-124)"""
    stage = (step + stage_steps) // (2 * stage_steps)
    progress = (step + stage_steps) / (2 * stage_steps) - stage
    return stage, progress


def result_path_of(flags, num_point, num_shape_point, model_dir):
    """main.py:392-414"""
    append_name = ["pWhole" if num_point is None else "p%d" % num_point,
                   "sWhole" if num_shape_point is None else "s%d" % num_shape_point]
    append_name += ["s{}".format("{:.4f}".format(flags.jitter_sigma).replace(".", ""))] if flags.jitter else ["clean"]
    if flags.drop_out < 1:
        append_name += ["d{}".format("{:.2f}".format(flags.drop_out).replace(".", ""))]
    return flags.result_dir or os.path.join(model_dir, 'result', 'x%d' % flags.up_ratio, "_".join(append_name))


def test(flags, net, device, num_point, result_dir):
    """upsample point clouds (main.py:333-389)"""
    if flags.ckpt != "random":
        pytorch_utils.load_network(net, flags.ckpt)
    net.to(device)
    net.eval()
    test_files = glob(flags.test_data, recursive=True)
    for point_path in test_files:
        folder = os.path.basename(os.path.dirname(point_path))
        out_path = os.path.join(result_dir, folder, point_path.split('/')[-1][:-4] + '.ply')
        data = pc_utils.load(point_path, flags.num_shape_point)[np.newaxis, ...]
        num_shape_point = data.shape[1] * flags.drop_out
        if flags.drop_out < 1:
            _, d = operations.furthest_point_sample(torch.from_numpy(data).to(device), int(num_shape_point),
                                                    NCHW=False)
            data = d.cpu().numpy()
        data, centroid, furthest_distance = pc_utils.normalize_point_cloud(data)
        is_2D = np.all(data[:, :, 2] == 0)
        if flags.jitter:
            data = pc_utils.jitter_perturbation_point_cloud(
                data, sigma=flags.jitter_sigma, clip=flags.jitter_max, is_2D=is_2D)
        data_t = torch.from_numpy(data.astype(np.float32)).transpose(2, 1).contiguous().to(device=device)
        print(os.path.basename(point_path))
        start = time.time()
        pred_pc = pipeline.upsample(net, data_t, num_point or data_t.shape[2], flags.up_ratio,
                                    flags.patch_num_ratio, final_fps=False)
        torch.cuda.synchronize(device)
        print("total time: ", time.time() - start)
        # main.py:379-380 (int(num_shape_point) * UP_RATIO output points)
        idx = operations.fps(pred_pc.transpose(2, 1).contiguous(), int(num_shape_point) * flags.up_ratio)
        pred = torch.gather(pred_pc.transpose(2, 1), 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
        pred = pred.cpu().numpy() * furthest_distance + centroid
        data_out = data * furthest_distance + centroid
        pc_utils.save_ply(data_out[0], out_path[:-4] + '_input.ply')
        pc_utils.save_ply(pred[0], out_path[:-4] + '.ply')


def synthetic_pairs(batch_size, num_point, ratio, device, seed):
    """Patch pairs with the array shapes H5Dataset.__getitem__ returns (data.py:174-196):
    input (B,3,num_point), label (B,3,num_point*ratio), normalised by the label patch."""
    g = torch.Generator().manual_seed(seed)
    lab = torch.randn(batch_size, num_point * ratio, 3, generator=g)
    lab = lab / lab.norm(dim=2, keepdim=True)
    lab = lab * torch.tensor([1.0, 1.0, 0.35]) + 0.02 * torch.randn(batch_size, num_point * ratio, 3, generator=g)
    sel = torch.stack([torch.randperm(num_point * ratio, generator=g)[:num_point] for _ in range(batch_size)])
    inp = torch.gather(lab, 1, sel.unsqueeze(-1).expand(-1, -1, 3))
    c = lab.mean(dim=1, keepdim=True)
    r = (lab - c).norm(dim=2).amax(dim=1).view(-1, 1, 1)
    return (((inp - c) / r).transpose(2, 1).contiguous().to(device),
            ((lab - c) / r).transpose(2, 1).contiguous().to(device))


def train(flags, net, device, num_point, model_dir):
    """main.py:127-211 without visdom.  `--h5_data <file>`: the reference's loop over H5Dataset
    (data.py; resident on the device, .npz or -- with h5py -- .hdf5) with its stage logic: a new
    ratio per stage, "combined" sampling after half a stage, Chamfer threshold after 60 %.
    `--h5_data synthetic`: the same schedule over generated Poisson-sphere pairs (no file needed)."""
    net.to(device)
    net.train()
    model = Model(net, "train", flags)
    steps_per_epoch = int(os.environ.get("TPU3_STEPS_PER_EPOCH", "0"))
    if flags.h5_data != "synthetic":
        H5Dataset = importlib.import_module(_pkg.__name__ + ".data").H5Dataset
        dataset = H5Dataset(h5_path=flags.h5_data, num_shape_point=flags.num_shape_point, num_patch_point=num_point,
                            batch_size=flags.batch_size, up_ratio=flags.up_ratio, step_ratio=flags.step_ratio,
                            device=device)
        steps_per_epoch = steps_per_epoch or len(dataset)
        start_epoch = model.step // steps_per_epoch
        stage, progress = get_stage_progress(model.step, flags.stage_steps)
        dataset.set_max_ratio(flags.step_ratio ** (stage + 1))
        if progress > 0.5:
            dataset.set_combined()
            if progress > 0.6:
                model.chamfer_criteria.set_threshold(flags.cd_threshold)
        else:
            model.chamfer_criteria.unset_threshold()
            dataset.unset_combined()
        for epoch in range(start_epoch + 1, flags.max_epoch):
            for i in range(steps_per_epoch):
                input_pc, label_pc, ratio = dataset[i]           # device tensors (B,3,M), (B,3,r*M)
                model.set_input(input_pc, ratio, label_pc=label_pc)
                model.optimize()
                new_stage, new_progress = get_stage_progress(model.step, flags.stage_steps)
                if stage + 1 == new_stage:                      # next stage: one more ratio
                    dataset.add_next_ratio()
                    dataset.unset_combined()
                    model.chamfer_criteria.unset_threshold()
                if progress <= 0.5 and new_progress > 0.5:
                    dataset.set_combined()
                if new_progress > 0.6:
                    model.chamfer_criteria.set_threshold(flags.cd_threshold)
                stage, progress = new_stage, new_progress
            print("epoch %d: " % epoch + ", ".join(["{}={}".format(k, v) for k, v in model.error_log.items()]))
            if epoch % 20 == 0:
                pytorch_utils.save_network(net, model_dir, "model", epoch_label=str(epoch), step=str(model.step))
        return
    steps_per_epoch = steps_per_epoch or 100
    start_epoch = model.step // steps_per_epoch
    stage, progress = get_stage_progress(model.step, flags.stage_steps)
    num_levels = net.num_levels
    for epoch in range(start_epoch + 1, flags.max_epoch):
        for i in range(steps_per_epoch):
            stage, progress = get_stage_progress(model.step, flags.stage_steps)
            max_level = min(stage + 1, num_levels)
            if progress > 0.5:       # "combined" stage: any ratio seen so far (data.py:63-76)
                level = 1 + (model.step % max_level)
            else:
                level = max_level
            ratio = flags.step_ratio ** level
            if progress > 0.6:
                model.chamfer_criteria.set_threshold(flags.cd_threshold)
            else:
                model.chamfer_criteria.unset_threshold()
            inp, lab = synthetic_pairs(flags.batch_size, num_point, ratio, device, model.step)
            model.set_input(inp, ratio, label_pc=lab)
            model.optimize()
        print("epoch %d: " % epoch + ", ".join(["{}={}".format(k, v) for k, v in model.error_log.items()]))
        if epoch % 20 == 0:
            pytorch_utils.save_network(net, model_dir, "model", epoch_label=str(epoch), step=str(model.step))


def main(argv=None):
    flags = build_parser().parse_args(argv)
    device = torch.device('cuda', flags.gpu)
    model_dir = os.path.join(flags.log_dir, flags.id)
    num_shape_point, num_point = flags.num_shape_point, flags.num_point
    assert(num_shape_point is not None or num_point is not None)
    num_point = num_point or int(num_shape_point * flags.drop_out)
    # main.py:114-115 (max_num_point is not forwarded: inner patches stay at 312 points)
    net = Net(max_up_ratio=flags.up_ratio, step_ratio=flags.step_ratio, knn=flags.knn,
              growth_rate=flags.growth_rate, dense_n=flags.dense_n, fm_knn=flags.fm_knn)
    result_dir = result_path_of(flags, flags.num_point, num_shape_point, model_dir)
    if flags.phase == "test":
        assert(flags.ckpt is not None)
        test(flags, net, device, num_point, result_dir)
    elif flags.phase == "train":
        ckpt = flags.ckpt
        train(flags, net, device, num_point, model_dir)
    elif flags.phase == "vis":
        raise SystemExit("--phase vis (interactive matplotlib viewer, main.py:288-330) is out of scope")
    else:
        raise SystemExit("unknown --phase %r" % flags.phase)


if __name__ == "__main__":
    main()
