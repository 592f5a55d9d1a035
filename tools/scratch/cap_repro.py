import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
main_side = torch.cuda.Stream()
side, aux, aux2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def body():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = a + 1
        ev = torch.cuda.Event(); ev.record(side)
    if mode in ("nested", "nested2", "nested_evjoin"):
        aux.wait_event(ev)
        with torch.cuda.stream(aux):
            c = b * 2
            evq = torch.cuda.Event(); evq.record(aux)
        with torch.cuda.stream(side):
            d = b + 3
            ev2 = torch.cuda.Event(); ev2.record(side)
        if mode == "nested2":
            aux.wait_event(ev2)
            with torch.cuda.stream(aux):
                c = c + d
        if mode == "nested_evjoin":
            e3 = torch.cuda.Event(); e3.record(aux); side.wait_event(e3)
        else:
            side.wait_stream(aux)
    main.wait_stream(side)
    return b
for _ in range(2):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(out[0]))
