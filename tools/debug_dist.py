import faulthandler, os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.dump_traceback_later(50, exit=True)
import test_distributed as T

def w(rank, world, port, mode, out):
    faulthandler.dump_traceback_later(40, exit=True)
    print("worker", rank, "start", flush=True)
    T._gpu_worker(rank, world, port, mode, out)
    print("worker", rank, "done", flush=True)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(w, args=(2, T._free_port(), sys.argv[1], "/tmp/o.npy"), nprocs=2, join=True)
    print("ok")
