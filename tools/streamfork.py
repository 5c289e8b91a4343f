"""What a cross-stream dependency costs on this stack: five tiny kernels per iteration on one stream, against the same
five with one of them forked to a second stream and joined again (two hipStreamWaitEvent edges per iteration).

    python tools/streamfork.py

MI355X / ROCm 7.2: 20.7 us serial vs 34.6 us forked per iteration -- about 7 us per edge, more than a kernel boundary
(4.1 us).  This is why the training step keeps every launch on one stream."""
import torch, time
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev); c = torch.zeros(1 << 20, device=dev)
def run(fork, n=2000):
    e1 = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        st.record(s1)
        for i in range(n):
            a.add_(1.0)                      # "L3"
            if fork:
                e1[0].record(s1)
                s2.wait_event(e1[0])
                with torch.cuda.stream(s2):
                    b.add_(1.0)              # "L4b" on the side stream
                    e1[1].record(s2)
                c.add_(1.0)                  # "L4a"
                c.mul_(1.0)                  # "L1"
                s1.wait_event(e1[1])
            else:
                b.add_(1.0); c.add_(1.0); c.mul_(1.0)
            a.mul_(1.0)                      # "L2"
        en.record(s1)
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
for r in range(3):
    print("serial %.2f us/iter   forked %.2f us/iter" % (run(False), run(True)))
