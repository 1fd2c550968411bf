"""Checkpoint wire format of the reference: what utils/misc.py:195-216 writes and
optimizer.py:45-89 reads -- {'state_dict' (keys carry DDP's 'module.' prefix), 'optimizer', 'scheduler', 'epoch', 'mean_iu',
'memory': m_items}. Files written by the reference load into the HIP model and vice versa."""
import torch


def _unwrap(net):
    return net.module if hasattr(net, 'module') else net


def snapshot_dict(net, optimizer=None, scheduler=None, epoch=0, mean_iu=0.0):
    """No collective in here: the reference saves on rank 0 only (train.py:188-191). With more than one rank and a memory commit still deferred
    (harness.agg_train_step), `m_items` raises instead of hanging the job -- harness.save_checkpoint finishes the commit on every rank first."""
    m = _unwrap(net)
    sd = {('module.' + k): v for k, v in m.state_dict().items()}      # the reference always saves the DDP-wrapped net
    out = {'state_dict': sd, 'epoch': epoch, 'mean_iu': mean_iu}
    if optimizer is not None:
        out['optimizer'] = optimizer.state_dict()
    if scheduler is not None:
        out['scheduler'] = scheduler.state_dict()
    if getattr(m, 'memory', None) is not None:
        out['memory'] = m.memory.m_items.detach()
    return out


def save_snapshot(path, net, optimizer=None, scheduler=None, epoch=0, mean_iu=0.0):
    torch.save(snapshot_dict(net, optimizer, scheduler, epoch, mean_iu), path)


def forgiving_state_restore(net, loaded_dict, verbose=False):
    """optimizer.py:73-89: copy every entry whose name and shape match, skip the rest. Accepts keys with or without the
    'module.' prefix whatever the wrapping of `net`; conv weights keep their channels-last (KRSC) memory."""
    m = _unwrap(net)
    if any(p.is_cuda for p in m.parameters()):
        from .hip import ops as _ops
        _ops.wait_commit()      # load_state_dict writes the weights a commit forward on its own stream may still be reading
    own = m.state_dict()
    new = {}
    for k in own:
        for cand in (k, 'module.' + k):
            if cand in loaded_dict and own[k].size() == loaded_dict[cand].size():
                new[k] = loaded_dict[cand]
                break
        else:
            if verbose:
                print('Do not match with saved parameter ', k)
    own.update(new)
    m.load_state_dict(own)
    return net, sorted(set(own) - set(new))


def restore_snapshot(net, optimizer, scheduler, snapshot, restore_optimizer_bool=False, map_location='cpu'):
    """optimizer.py:45-70. Returns (net, optimizer, scheduler, epoch, mean_iu)."""
    ckpt = torch.load(snapshot, map_location=map_location) if isinstance(snapshot, str) else snapshot
    if optimizer is not None and 'optimizer' in ckpt and restore_optimizer_bool:
        optimizer.load_state_dict(ckpt['optimizer'])
    if scheduler is not None and 'scheduler' in ckpt and restore_optimizer_bool:
        scheduler.load_state_dict(ckpt['scheduler'])
    forgiving_state_restore(net, ckpt['state_dict'] if 'state_dict' in ckpt else ckpt)
    m = _unwrap(net)
    if 'memory' in ckpt and getattr(m, 'memory', None) is not None:
        if getattr(m.memory, 'commit_owed', False):
            # a restore on a SUBSET of ranks would silently drop the owed all-reduce on those ranks only and leave the others waiting in theirs
            raise RuntimeError('restore_snapshot: the memory-slot all-reduce of the last commit forward is still pending. Call '
                               'pinthememory_amd.harness.finish_commit(net) on EVERY rank before restoring a snapshot between steps.')
        dev = next(m.parameters()).device
        m.memory.m_items = ckpt['memory'].to(dev)
    return net, optimizer, scheduler, ckpt.get('epoch', 0), ckpt.get('mean_iu', 0.0)
