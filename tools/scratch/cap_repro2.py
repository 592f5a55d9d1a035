import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
side, aux = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    if mode in ("origin_fork", "origin_fork_join_side"):
        aux.wait_stream(main)
    with torch.cuda.stream(side):
        b = a + 1
        ev = torch.cuda.Event(); ev.record(side)
    aux.wait_event(ev)
    with torch.cuda.stream(aux):
        c = b * 2
    with torch.cuda.stream(side):
        d = b + 3
    if mode in ("join_main", "origin_fork"):
        main.wait_stream(aux)
    else:
        side.wait_stream(aux)
    main.wait_stream(side)
    return c
for _ in range(2):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(out[0]))
