import torch
def graph_time(fn, per_graph=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(per_graph): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9; tot = 0.0
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); e.synchronize()
        t = s.elapsed_time(e) * 1e3 / per_graph
        best = min(best, t); tot += t
    return best, tot / reps
