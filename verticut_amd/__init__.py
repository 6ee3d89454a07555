"""verticut_amd -- MI355X (gfx950) Hamming k-NN engine behind VertiCut's search interfaces.

    engine.py   ctypes binding of include/verticut_gpu.h (libverticut_gpu.so, built by `python -m verticut_amd.build`)
    sharded.py  database sharded over the GPUs of a node (torch.distributed / RCCL all-gather of per-shard top-k)
    csrc/       HIP kernels + C ABI;  host/  C++ layer with the reference's class shapes and drivers

Nothing here computes on the CPU: without the built library and a gfx950 device every entry point raises.
"""
__version__ = "0.1.0"
