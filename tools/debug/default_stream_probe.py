"""Does work on a torch pool stream run beside work on the DEFAULT stream on this runtime?  A 0.2 ms spin on the main
stream, a short spin on a side stream enqueued behind the spin's START: when does the side stream's kernel finish?
main = the default stream, then main = a pool stream.  usage: python tools/debug/default_stream_probe.py"""
import torch

dev = torch.device("cuda", 0)


def probe(main, label):
    for k in range(4):
        c = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        e0, e_main, e_c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(main):
            e0.record()
            torch.cuda._sleep(400_000)
            e_main.record()
        c.wait_event(e0)
        with torch.cuda.stream(c):
            torch.cuda._sleep(2_000)
            e_c.record()
        torch.cuda.synchronize()
        print(f"{label}: side stream {k} done after {e0.elapsed_time(e_c) * 1e3:6.0f} us, main's spin after {e0.elapsed_time(e_main) * 1e3:6.0f} us")


torch.cuda._sleep(1000)
torch.cuda.synchronize()
probe(torch.cuda.default_stream(dev), "main = default stream")
probe(torch.cuda.Stream(device=dev), "main = pool stream   ")
