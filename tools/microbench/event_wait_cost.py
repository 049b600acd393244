"""Cost of hipStreamWaitEvent on MI355X: a wait on an event that has ALREADY completed inserts nothing into the stream, a wait on
a fresh event costs a barrier packet (why vloam_process_scan enqueues the odometry one sweep late).
Output of one run: profiles/r01_event_wait_cost.txt."""
import torch, time
dev = torch.device('cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
def run(mode, n=200):
    # mode: 'none' no wait; 'done' wait on long-completed event; 'fresh' wait on event recorded just now on s1 after a tiny op
    e_done = torch.cuda.Event(); 
    with torch.cuda.stream(s1):
        x.add_(1); e_done.record()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s2):
        st.record()
        for i in range(n):
            y.add_(1)
            if mode == 'done':
                s2.wait_event(e_done)
            elif mode == 'fresh':
                e = torch.cuda.Event()
                with torch.cuda.stream(s1):
                    x.add_(1); e.record()
                s2.wait_event(e)
            y.add_(1)
        en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / n
for mode in ('none', 'done', 'fresh', 'none', 'done', 'fresh'):
    print(mode, 'us per (op, [wait], op):', round(run(mode), 2))
