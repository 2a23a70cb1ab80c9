"""
Independent mappings side by side on one GPU.

The reference runs independent mappings strictly one after the other: `cross_val` trains one mapping per held-out gene
(utils.py:576-600; 249 in the tutorial), the tuning driver three seeds per trial (mapping_parameter_tuning.py:109-131).
Clusters-mode problems are tiny (18 x 250 x 9852) and latency-bound on a 256-CU GPU: one iteration is ~7 dependent
kernels of ~10 us that each occupy a fraction of the chip.  `train_many` gives every mapping its own HIP stream and its
own host thread (the C ABI releases the GIL for the whole `tg_mapper_step` loop, and different handles may be driven from
different threads), so the kernels of different mappings fill the idle CUs.  Results are bit-identical to training the
same mappings one by one: nothing is shared between handles.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import torch


def train_many(builders, num_epochs, learning_rate=0.1, max_concurrent=4, device="cuda:0", **train_kwargs):
    """Train independent mappings concurrently.

    builders: callables, each returning a `Mapper` / `MapperConstrained`.  They are called one after the other on the
              calling thread (the reference's initialisation draws from the global NumPy RNG, `np.random.seed(random_state)`,
              which must not be interleaved), each under its own HIP stream so that the mapper binds to it; only the
              training loops run concurrently.
    Returns the list of `mapper.train(...)` results (in the order of `builders`) and the mappers themselves."""
    device = torch.device(device)
    builders = list(builders)
    n = len(builders)
    results, mappers, streams = [None] * n, [None] * n, [None] * n
    with torch.cuda.device(device):
        for i in range(n):
            streams[i] = torch.cuda.Stream(device=device)
            with torch.cuda.stream(streams[i]):
                mappers[i] = builders[i]()
            streams[i].synchronize()

    def work(i):
        with torch.cuda.device(device), torch.cuda.stream(streams[i]):
            results[i] = mappers[i].train(num_epochs=num_epochs, learning_rate=learning_rate, print_each=None, **train_kwargs)
        streams[i].synchronize()

    with ThreadPoolExecutor(max_workers=max(1, int(max_concurrent))) as pool:
        for f in [pool.submit(work, i) for i in range(n)]:
            f.result()
    return results, mappers
