"""Graph-replayed kernel timing shared by the developer scripts."""
import torch


def time_graph(fn_list, iters=20):
    """fn_list: callables each enqueuing one kernel.  Returns mean us per kernel, graph-replayed."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fn_list:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fn_list:
                f()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * len(fn_list))
