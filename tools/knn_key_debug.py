import importlib, sys, torch
sys.path.insert(0, ".")
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(2, 312, 24, device=dev)
be = ops.BACKEND
be.graph_dup_events(reset=True)
opt = be.knn_graph(33, x, optimistic=True)
print("events", be.graph_dup_events(reset=True))
ex = be.knn_graph(33, x, optimistic=False)
d = torch.cdist(x, x) ** 2
ref = d.topk(33, dim=-1, largest=False).indices
for q in (0, 5, 311):
    print("q", q, "opt ", sorted(opt[0, q].tolist())[:12], "slot0", int(opt[0, q, 0]))
    print("q", q, "ex  ", sorted(ex[0, q].tolist())[:12], "slot0", int(ex[0, q, 0]))
    print("q", q, "ref ", sorted(ref[0, q].tolist())[:12])
same = (opt.sort(-1)[0] == ex.sort(-1)[0]).all(-1).float().mean()
print("sets equal fraction", float(same))
cnt_self = (opt[:, :, 1:] == torch.arange(312, device=dev).view(1, -1, 1)).sum()
print("self appears among the others:", int(cnt_self))
