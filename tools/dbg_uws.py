import sys, os, ctypes, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("3pu_pytorch_amd._lib")
lib = L.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (b, n, c, k) in [(64, 312, 24, 33), (8, 3120, 3, 5)]:
    p = torch.randn(b, n, c, device=dev)
    q = p[:, :312].contiguous()
    m = q.shape[1]
    dup = torch.zeros(b, n, dtype=torch.uint8, device=dev)
    uws = torch.full((8,), 77, dtype=torch.int32, device=dev)
    idx = torch.zeros(b, m, k, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    print("prepare rc", lib.tpu3_knn_unique_prepare_f32(s, b, m, n, c, q.data_ptr(), p.data_ptr(), None, dup.data_ptr(), uws.data_ptr(), None, 0))
    torch.cuda.synchronize(); print("uws after prepare", uws.tolist(), "dup sum", int(dup.sum()))
    print("knn rc", lib.tpu3_knn_f32(s, b, m, n, c, k, q.data_ptr(), p.data_ptr(), None, dup.data_ptr(), uws.data_ptr(), idx.data_ptr(), 8, None, None))
    torch.cuda.synchronize(); print("uws after knn", uws.tolist())
