"""Micro-benchmark: cost of cross-stream dependencies inside a captured hipGraph.  Two chains, A (short, ~10 us) and B
(long, bandwidth-bound, ~60 us), with edges A_t -> B_t and B_{t-1} -> A_{t+1} (the shape of k_step(t+1) || k_map(t));
compared with the same kernels serialised on one stream."""
import torch, time
dev = torch.device("cuda")
T = 100
src = torch.empty(80 * 1024 * 1024 // 4, device=dev)         # 80 MB read + 240 MB written ~ k_map's traffic
dst = [torch.empty_like(src) for _ in range(3)]
small = torch.zeros(512 * 1024, device=dev)

def A():
    for _ in range(3):
        small.add_(1.0)          # a few dependent tiny kernels ~ latency-bound k_step

def B():
    for d in dst:
        d.copy_(src)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / T * 1e3

# serial graph
g1 = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    A(); B(); torch.cuda.synchronize()
    with torch.cuda.graph(g1, stream=s):
        for t in range(T):
            A(); B()
print(f"serial graph      : {timed(g1.replay):7.2f} us / iteration")
gA = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(gA, stream=s):
        for t in range(T): A()
gB = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(gB, stream=s):
        for t in range(T): B()
print(f"A alone           : {timed(gA.replay):7.2f} us / iteration;  B alone: {timed(gB.replay):7.2f}")

# pipelined graph: A on a side stream, B on the capture stream
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s):
        evA = [torch.cuda.Event() for _ in range(T + 1)]
        evB = [torch.cuda.Event() for _ in range(T + 1)]
        side.wait_stream(s)
        with torch.cuda.stream(side):
            A(); evA[0].record(side)
        for t in range(T):
            s.wait_event(evA[t])           # B_t needs A_t
            if t + 1 < T:
                with torch.cuda.stream(side):
                    if t >= 1:
                        side.wait_event(evB[t - 1])   # A_{t+1} reuses the buffer B_{t-1} read
                    A(); evA[t + 1].record(side)
            B(); evB[t].record(s)
        s.wait_stream(side)
print(f"pipelined graph   : {timed(g2.replay):7.2f} us / iteration")

# pipelined, eager (no graph)
def eager():
    evA = [torch.cuda.Event() for _ in range(T + 1)]
    evB = [torch.cuda.Event() for _ in range(T + 1)]
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        A(); evA[0].record(side)
    for t in range(T):
        cur.wait_event(evA[t])
        if t + 1 < T:
            with torch.cuda.stream(side):
                if t >= 1:
                    side.wait_event(evB[t - 1])
                A(); evA[t + 1].record(side)
        B(); evB[t].record(cur)
    cur.wait_stream(side)
print(f"pipelined eager   : {timed(eager):7.2f} us / iteration")
def serial_eager():
    for t in range(T): A(); B()
print(f"serial eager      : {timed(serial_eager):7.2f} us / iteration")
