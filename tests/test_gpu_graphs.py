"""hipGraph capture of whole model steps (poisson_cnn_amd/graphs.py): a replayed step is the same kernels on the same buffers as the eager
step, so results must be bit-identical - inference and training (loss, gradients, weights after two optimizer steps) of the boundary
network, the end-to-end model and the homogeneous network; a shape other than the captured one is refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dbcnn(seed):
    from oracle import dbcnn as odb
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.models import Dirichlet_BC_NN_Legacy_2
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.dbcnn_tiny()
    m = Dirichlet_BC_NN_Legacy_2(**full['model'])
    m.set_weights(odb.init_params(full['model'], seed=seed, gain=1.3, randomize_all=True))
    m.compile(loss=loss_wrapper(global_batch_size=3, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
    return m


def _dbcnn_batch(seed):
    g = torch.Generator().manual_seed(seed)
    bc = torch.cumsum(torch.randn(3, 1, 44, generator=g) * 0.1, 2).cuda()
    dx = (torch.rand(3, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
    tgt = (torch.randn(3, 1, 40, 44, generator=g) * 0.1).cuda()
    return bc, dx, tgt


def test_dbcnn_graph_replay_is_bit_identical_to_eager():
    from poisson_cnn_amd.graphs import GraphedInference, GraphedTrainStep
    eager, graphed = _dbcnn(5), _dbcnn(5)
    bc, dx, tgt = _dbcnn_batch(1)
    inf = GraphedInference(graphed, [bc, dx, 40])
    bc2, dx2, tgt2 = _dbcnn_batch(2)
    assert torch.equal(inf([bc2, dx2, 40]), eager([bc2, dx2, 40]))
    assert torch.equal(inf([bc, dx, 40]), eager([bc, dx, 40]))
    with pytest.raises(ValueError, match='captured for'):
        inf([bc, dx, 36])
    step = GraphedTrainStep(graphed, ((bc, dx), tgt))
    assert torch.equal(eager.store.flat_w, graphed.store.flat_w)                  # capture leaves the weights alone
    for data in (((bc, dx), tgt), ((bc2, dx2), tgt2)):
        le = eager.train_step(data)
        lg = step(data)
        assert float(le['loss']) == float(lg['loss']) and float(le['mse']) == float(lg['mse'])
        assert torch.equal(eager.store.flat_g, graphed.store.flat_g)
        assert torch.equal(eager.store.flat_w, graphed.store.flat_w)
    assert graphed.optimizer.iterations == 2


def test_hpnn_graph_replay_is_bit_identical_to_eager():
    from oracle import hpnn as ohpnn
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.graphs import GraphedTrainStep
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn_tiny()
    models = []
    for _ in range(2):
        m = Homogeneous_Poisson_NN_Legacy(**full['model'])
        m.set_weights(ohpnn.init_params(full['model'], seed=3, gain=1.4, randomize_all=True))
        m.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
        models.append(m)
    eager, graphed = models
    rng = np.random.default_rng(7)
    rhs = torch.tensor(rng.uniform(-1, 1, (2, 1, 40, 44)).astype(np.float32)).cuda()
    dx = torch.tensor(rng.uniform(5e-3, 5e-2, (2, 1)).astype(np.float32)).cuda()
    tgt = torch.tensor((rng.standard_normal((2, 1, 40, 44)) * 0.1).astype(np.float32)).cuda()
    import os
    step = GraphedTrainStep(graphed, ((rhs, dx), tgt))
    eager.ctx.use_side = False                                                     # same single-stream launch order as the captured step
    for _ in range(2):
        le, lg = eager.train_step(((rhs, dx), tgt)), step(((rhs, dx), tgt))
        assert float(le['loss']) == float(lg['loss'])
        assert torch.equal(eager.store.flat_w, graphed.store.flat_w)


def test_an_earlier_capture_survives_a_later_capture_of_a_larger_shape():
    """a captured graph bakes in the addresses of model-level scratch (Context.ws / ws_side / flipped-filter scratch,
    ops._default_ws) and of the capture stream's handle-owned workspaces; a later capture or eager step on a LARGER shape makes their owners
    allocate bigger buffers.  The earlier graph must keep replaying into valid memory: capture A (small), capture B (larger, same model),
    run eager steps on B's shape, empty the allocator's cache, then replay A and compare with an eager twin - bit-identical."""
    from poisson_cnn_amd.graphs import GraphedInference, GraphedTrainStep

    def batch(seed, L, H):
        g = torch.Generator().manual_seed(seed)
        bc = torch.cumsum(torch.randn(3, 1, L, generator=g) * 0.1, 2).cuda()
        dx = (torch.rand(3, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
        tgt = (torch.randn(3, 1, H, L, generator=g) * 0.1).cuda()
        return bc, dx, tgt
    eager, graphed = _dbcnn(9), _dbcnn(9)
    bcA, dxA, tgtA = batch(1, 44, 40)
    bcB, dxB, tgtB = batch(2, 96, 88)
    infA = GraphedInference(graphed, [bcA, dxA, 40])
    stepA = GraphedTrainStep(graphed, ((bcA, dxA), tgtA))
    ptrs = [b.data_ptr() for b in stepA._keep]
    side = [(c.side_allowed, c.use_side) for c in [graphed.ctx]]
    infB = GraphedInference(graphed, [bcB, dxB, 88])                  # larger shape: every scratch owner regrows
    stepB = GraphedTrainStep(graphed, ((bcB, dxB), tgtB))
    assert torch.equal(infB([bcB, dxB, 88]), eager([bcB, dxB, 88]))
    for _ in range(2):                                                # eager steps on the larger shape as well
        graphed.train_step(((bcB, dxB), tgtB)); eager.train_step(((bcB, dxB), tgtB))
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                          # anything the owners dropped and nobody else holds goes back to the driver
    assert [b.data_ptr() for b in stepA._keep] == ptrs                # A's scratch is still A's
    assert torch.equal(infA([bcA, dxA, 40]), eager([bcA, dxA, 40]))
    le, lg = eager.train_step(((bcA, dxA), tgtA)), stepA(((bcA, dxA), tgtA))
    assert float(le['loss']) == float(lg['loss']) and torch.equal(eager.store.flat_g, graphed.store.flat_g) and torch.equal(eager.store.flat_w, graphed.store.flat_w)
    le, lg = eager.train_step(((bcB, dxB), tgtB)), stepB(((bcB, dxB), tgtB))
    assert float(le['loss']) == float(lg['loss']) and torch.equal(eager.store.flat_w, graphed.store.flat_w)
    assert [(c.side_allowed, c.use_side) for c in [graphed.ctx]] == side     # capture restores the side-stream settings


def test_graphed_step_keeps_collectives_and_the_learning_rate_out_of_the_graph():
    """with data parallelism attached, train_step's metric all-reduce (model.metric_sync) must not be recorded in the graph;
    it runs eagerly after each replay, like the gradient all-reduce, and logs['lr'] is the optimizer's current rate."""
    from poisson_cnn_amd.graphs import GraphedTrainStep
    m = _dbcnn(4)
    bc, dx, tgt = _dbcnn_batch(3)
    calls = {'grad': 0, 'metric': 0, 'capturing': False, 'in_capture': 0}

    def grad_sync(flat):
        calls['grad'] += 1
        calls['in_capture'] += torch.cuda.is_current_stream_capturing()
        return flat

    def metric_sync(loss, mse):
        calls['metric'] += 1
        calls['in_capture'] += torch.cuda.is_current_stream_capturing()
        return loss * 2, mse * 2                                       # a stand-in for the 2-rank sum
    m.grad_sync, m.metric_sync = grad_sync, metric_sync
    step = GraphedTrainStep(m, ((bc, dx), tgt))
    assert calls['grad'] == 0 and calls['metric'] == 0                 # neither ran during warm-up or capture
    m.optimizer.learning_rate = 5e-4
    logs = step(((bc, dx), tgt))
    assert calls == {'grad': 1, 'metric': 1, 'capturing': False, 'in_capture': 0}
    assert float(logs['loss']) == 2 * float(step.logs['loss']) and logs['lr'] == 5e-4
    assert m.grad_sync is grad_sync and m.metric_sync is metric_sync


def test_closing_captured_graphs_returns_their_handles_and_scratch():
    """every capture creates a stream, a libpcnn handle with pcnn_set_workspace_retain(1) and per-stream scratch entries in the model's
    contexts.  close() (or dropping the object) must give all of that back: the handle cache and the contexts' per-stream tables do not grow with
    the number of captures made, and a closed object refuses to replay."""
    import gc
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.graphs import GraphedInference, _ctxs
    m = _dbcnn(7)
    bc, dx, _ = _dbcnn_batch(3)
    ref = m([bc, dx, 40]).clone()
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    for i in range(4):
        inf = GraphedInference(m, [bc, dx, 40])
        sp = inf.stream.cuda_stream
        assert torch.equal(inf([bc, dx, 40]), ref)
        assert (dev, sp) in ops._handles                             # the capture stream has its own handle (and, for models that use them, scratch entries)
        if i % 2 == 0:
            inf.close()
            with pytest.raises(RuntimeError, match='closed'):
                inf([bc, dx, 40])
            inf.close()                                              # idempotent
        else:
            del inf                                                  # the finaliser closes it
            gc.collect()
        assert (dev, sp) not in ops._handles                         # ... and both are gone afterwards (a recycled stream pointer starts afresh)
        assert not any(sp in c._ws or sp in c._wflips for c in _ctxs(m))
    assert torch.equal(m([bc, dx, 40]), ref)                         # the eager path is untouched


def test_replay_survives_another_model_replacing_the_flipped_filter_table():
    """a captured train step records the launch that re-forms every layer's flipped filter from a device table of pointers
    (ops.sync_flipped_filters).  When another model's first backward changes the set of layers, the module-level table is rebuilt and the old tensor used to
    go back to the allocator - a later replay then read recycled memory as pointer entries.  The graph now owns the table (and the flipped filters) it
    recorded: capture A, let a second model B register its layers and die, churn the allocator, replay A - bit-identical to the eager twin."""
    import gc
    eager, graphed = _dbcnn(5), _dbcnn(5)
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.graphs import GraphedTrainStep
    bc, dx, tgt = _dbcnn_batch(1)
    step = GraphedTrainStep(graphed, ((bc, dx), tgt))
    table_at_capture = ops._flip_table[1]
    assert table_at_capture is not None and any(t is table_at_capture for t in step._keep)      # the graph holds what it recorded
    le, lg = eager.train_step(((bc, dx), tgt)), step(((bc, dx), tgt))
    assert float(le['loss']) == float(lg['loss']) and torch.equal(eager.store.flat_w, graphed.store.flat_w)
    other = _dbcnn(9)                                                               # a second model: its first backward joins the layer set -> a new table
    other.train_step(((bc, dx), tgt))
    assert ops._flip_table[1] is not table_at_capture
    del other
    gc.collect()
    junk = [torch.full((1 << 18,), float('nan'), device='cuda') for _ in range(64)]     # whatever was freed is overwritten with NaN bit patterns
    del junk
    torch.cuda.synchronize()
    bc2, dx2, tgt2 = _dbcnn_batch(2)
    for data in (((bc2, dx2), tgt2), ((bc, dx), tgt)):
        le, lg = eager.train_step(data), step(data)
        assert float(le['loss']) == float(lg['loss'])
        assert torch.equal(eager.store.flat_g, graphed.store.flat_g) and torch.equal(eager.store.flat_w, graphed.store.flat_w)
    step.close()
