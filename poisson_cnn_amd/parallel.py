"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Replaces tf.distribute.MirroredStrategy(cross_device_ops=ReductionToOneDevice()) (train/hpnn_legacy_train.py:37-38):
weights are replicated (identical seed, then a broadcast from rank 0), every global batch is split evenly by sample,
each rank's loss is already divided by the GLOBAL batch size (losses/loss_wrapper.py:46-49), so ONE all-reduce(SUM) of the
flat fp32 gradient bucket (5.56 M floats = 22.2 MB for hpnn.json) per optimizer step gives the mean gradient, followed by
the identical Adam step on every rank.  With BatchNormalization in TRAINING mode (`batchnorm_training=True`, not the reference's default) a second,
tiny all-reduce(MEAN) keeps the moving statistics equal on all ranks (`sync_bn_stats`).  No other collective is on the data path.
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, rank=0, world_size=1, local_rank=0, backend=None):
        self.rank, self.world_size, self.local_rank, self.backend = rank, world_size, local_rank, backend

    @classmethod
    def from_env(cls, backend=None, timeout_s=None):
        """timeout_s (or PCNN_DIST_TIMEOUT_S): how long a rank waits in the rendezvous / a collective before the process group gives up -
        a rank that died before a barrier then costs the others this long, not the job's whole time limit."""
        ws = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        lr = int(os.environ.get('LOCAL_RANK', '0'))
        if torch.cuda.is_available():
            torch.cuda.set_device(lr % max(torch.cuda.device_count(), 1))
        if ws > 1 and not dist.is_initialized():
            backend = backend or os.environ.get('PCNN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            import datetime
            timeout_s = float(timeout_s if timeout_s is not None else os.environ.get('PCNN_DIST_TIMEOUT_S', 1800))
            if backend == 'nccl':            # the RCCL watchdog aborts a collective that exceeds the timeout instead of hanging in it
                os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
            dist.init_process_group(backend=backend, rank=rank, world_size=ws, timeout=datetime.timedelta(seconds=timeout_s))
        if backend is None and dist.is_initialized():          # the caller's own process group: report and rendezvous over ITS backend
            backend = dist.get_backend()
        self = cls(rank, ws, lr, backend)
        if os.environ.get('PCNN_COLLECTIVE') == 'c_abi' and torch.cuda.is_available():
            self.enable_c_abi_collective()
        return self

    def local_batch(self, global_batch):
        """Even split by sample; the reference requires the same (H,W) on all replicas per step, and so do we."""
        if global_batch % self.world_size != 0:
            raise ValueError('global batch %d is not divisible by %d ranks' % (global_batch, self.world_size))
        return global_batch // self.world_size

    def shard(self, t):
        """This rank's slice of a global-batch tensor (dim 0)."""
        n = self.local_batch(t.shape[0])
        return t[self.rank * n:(self.rank + 1) * n]

    def all_reduce_sum(self, flat):
        if getattr(self, '_c_abi', None) is not None:
            import ctypes
            h = self._c_abi
            if torch.cuda.current_stream().cuda_stream != h.stream_ptr:
                raise RuntimeError('the C-ABI collective was initialised on another stream than the one the gradients are produced on')
            if flat.dtype != torch.float32 or not flat.is_contiguous():
                raise ValueError('pcnn_allreduce takes a contiguous fp32 buffer')
            h.call('pcnn_allreduce', ctypes.c_void_p(flat.data_ptr()), ctypes.c_size_t(flat.numel()))
        elif self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def enable_c_abi_collective(self):
        """Route the gradient all-reduce through the library's own entry point (pcnn_allreduce: RCCL bound inside libpcnn, issued on the
        handle's stream) instead of torch.distributed - the path a non-torch host program uses.  torch.distributed, if
        initialised, only ships the 128-byte rendezvous id from rank 0.  Also selected by PCNN_COLLECTIVE=c_abi in from_env()."""
        import ctypes
        from . import ops
        h = ops.handle()
        ident = self.exchange_unique_id(h)
        raw = (ctypes.c_ubyte * 128)(*ident)
        h.call('pcnn_comm_init', raw, ctypes.c_int(self.rank), ctypes.c_int(self.world_size))
        self._c_abi = h
        return self

    def exchange_unique_id(self, h=None):
        """The rendezvous of the C-ABI collective: rank 0 asks the library for RCCL's 128-byte unique id (pcnn_comm_unique_id) and ships it to every
        rank over the process group that already exists (gloo or nccl); returns the 128 bytes every rank then hands to pcnn_comm_init."""
        import ctypes
        from . import ops
        h = h or ops.handle()
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            h.call('pcnn_comm_unique_id', buf)
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if self.world_size > 1:
            dev = 'cuda' if self.backend == 'nccl' else 'cpu'
            ident = ident.to(dev)
            dist.broadcast(ident, src=0)
            ident = ident.cpu()
        return ident.tolist()

    def broadcast(self, flat, src=0):
        if self.world_size > 1:
            dist.broadcast(flat, src=src)
        return flat

    def attach(self, model):
        """Replicate rank 0's weights and hook the gradient all-reduce in front of the optimizer step."""
        for store in (model.stores if hasattr(model, 'stores') else [model.store]):
            self.broadcast(store.flat_w)
            self.broadcast(store.flat_stats)
        from . import ops
        ops.weights_changed()                                           # rank 0's weights have replaced this rank's: cached filter spectra are stale
        stores = list(model.stores if hasattr(model, 'stores') else [model.store])

        def grad_sync(flat_g):
            """Called by train_step with a store's flat gradient bucket in front of the optimizer step: all-reduce(SUM) of the gradients and -
            with BatchNormalization in training mode - all-reduce(MEAN) of that store's moving statistics, which each rank has just updated
            from ITS shard of the batch."""
            self.all_reduce_sum(flat_g)
            for st in stores:
                if getattr(st, 'flat_g', None) is flat_g:
                    self.sync_bn_stats(st)
            return flat_g
        model.grad_sync = grad_sync
        model.metric_sync = self.global_metrics
        return model

    def sync_bn_stats(self, store):
        """BatchNormalization moving mean / variance under data parallelism.  The reference's variables are MirroredVariables with
        aggregation = MEAN (models/Homogeneous_Poisson_NN_Legacy.py:53-57 builds the layers under the strategy scope of
        train/hpnn_legacy_train.py:37-41): after a training-mode step every replica holds the mean over the replicas of the per-replica
        updates (an all-reduce of 2 x 1 076 floats for hpnn.json).  Only when the statistics move at all (`batchnorm_training=True`;
        the default is the reference's inference-mode BN, whose statistics never change) - otherwise no collective is issued."""
        if self.world_size == 1 or not getattr(store, 'bn_training', False) or not getattr(store, 'nbn', 1):
            return
        stats = store.flat_stats
        if getattr(self, '_c_abi', None) is not None:
            self.all_reduce_sum(stats)
        else:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        stats.mul_(1.0 / self.world_size)

    def global_metrics(self, loss, mse):
        """Per-rank (loss share, local mse) -> (global loss, global mean mse) on every rank: a 2-element all-reduce(SUM).  Callbacks that
        act on these (learning-rate schedule, NaN termination, best-checkpoint) then take identical decisions on all ranks."""
        if self.world_size == 1:
            return loss, mse
        t = torch.stack([torch.as_tensor(loss, dtype=torch.float32).reshape(()), torch.as_tensor(mse, dtype=torch.float32).reshape(()) / self.world_size])
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t[0], t[1]

    def collective_name(self):
        """The collective actually in use, for reports: backend "nccl" is RCCL on ROCm."""
        if getattr(self, '_c_abi', None) is not None:
            return 'RCCL all-reduce (pcnn_allreduce, C-ABI)'
        if self.world_size == 1:
            return 'none (single rank)'
        return {'nccl': 'RCCL all-reduce', 'gloo': 'gloo all-reduce (CPU)'}.get(self.backend, '%s all-reduce' % self.backend)

    def ranks_seen(self):
        """World size as the process group itself reports it (1 without a group) - for reports: proof that N ranks really joined."""
        return dist.get_world_size() if dist.is_initialized() else 1

    def rccl_version(self):
        """RCCL's version when the nccl backend carries the collectives, else None."""
        if self.backend != 'nccl':
            return None
        try:
            return '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001 - a report field must not kill the run
            return 'unknown (%r)' % (e,)

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()

    def max_over_ranks(self, value):
        if self.world_size == 1:
            return value
        dev = 'cuda' if (self.backend == 'nccl') else 'cpu'
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_over_ranks(self, value):
        """Every rank's scalar, in rank order, on every rank (reports: a slow rank must be distinguishable from collective cost)."""
        if self.world_size == 1:
            return [float(value)]
        dev = 'cuda' if (self.backend == 'nccl') else 'cpu'
        t = torch.zeros(self.world_size, dtype=torch.float64, device=dev)
        t[self.rank] = float(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.cpu().tolist()]

