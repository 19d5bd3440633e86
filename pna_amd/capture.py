"""hipGraph capture of a layer forward (inference).

For batched small graphs (ZINC: ~3 k nodes / 6 k edges per batch) the layer is launch-latency bound: a tower
layer issues ~25 kernels of a few microseconds each.  Every pna_amd kernel is launched on torch's current
stream through the C ABI, so the whole forward can be recorded into one hipGraph with torch.cuda.CUDAGraph and
replayed with a single launch.  Everything that synchronises (CSR construction, heavy-row schedule, degree
scalers, weight packing) happens in the warm-up calls before the capture.
"""
import torch


class GraphedForward:
    """callable(*tensor_args) that replays a captured `fn(*static_args)`.

    `fn` must be a pure inference function of its tensor arguments (shapes fixed); non-tensor arguments are
    bound at construction.  The static input buffers are owned by this object; call() copies into them.
    """

    def __init__(self, fn, *example_args, warmup=3, alias_inputs=False):
        # static inputs keep the example's strides (a (V, 75) view of an 80-float pitch stays one: the one-kernel layer reads
        # 16-byte aligned rows); alias_inputs=True records the example tensors themselves (no per-call copy: the caller updates
        # them in place)
        def static(a):
            if not torch.is_tensor(a) or alias_inputs:
                return a
            b = torch.empty_strided(a.shape, a.stride(), dtype=a.dtype, device=a.device)
            b.copy_(a)
            return b
        self.static_in = [static(a) for a in example_args]
        dev = next(a.device for a in example_args if torch.is_tensor(a))
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *args):
        for dst, src in zip(self.static_in, args):
            if torch.is_tensor(dst) and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_out
