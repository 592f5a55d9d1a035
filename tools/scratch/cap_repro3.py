import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
side, aux, aux2, fin = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def body():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    if mode == "fin":
        fin.wait_stream(main)
    with torch.cuda.stream(side):
        b = a + 1
        ev = torch.cuda.Event(); ev.record(side)
    aux.wait_event(ev)
    with torch.cuda.stream(aux):
        c = b * 2
    with torch.cuda.stream(side):
        d = b + 3
        ev2 = torch.cuda.Event(); ev2.record(side)
    aux2.wait_event(ev2)
    with torch.cuda.stream(aux2):
        e = d * 2
    if mode == "second_wait":
        aux.wait_event(ev2)
        with torch.cuda.stream(aux):
            c = c + d
        main.wait_stream(aux); main.wait_stream(aux2); main.wait_stream(side)
    elif mode == "sibling":
        aux.wait_stream(aux2)
        with torch.cuda.stream(aux):
            c = c + e
        main.wait_stream(aux); main.wait_stream(side)
    elif mode == "fin":
        fin.wait_stream(side); fin.wait_stream(aux); fin.wait_stream(aux2)
        with torch.cuda.stream(fin):
            c = c + e + d
        main.wait_stream(fin)
    elif mode == "fin_late":   # fin forks from side at the end
        fin.wait_stream(side); fin.wait_stream(aux); fin.wait_stream(aux2)
        with torch.cuda.stream(fin):
            c = c + e + d
        main.wait_stream(fin)
    return c
for _ in range(2):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(out[0]))
