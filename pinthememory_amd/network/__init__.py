"""Drop-in boundary: same entry points as /root/reference/network/__init__.py (get_net :12-22,
warp_network_in_dataparallel :25-33, get_model :36-46). `args.arch` strings resolve through importlib exactly as in the
reference, e.g. 'pinthememory_amd.network.deepv3plus.DeepR50V3PlusD' (or 'network.deepv3plus.DeepR50V3PlusD' when this
package is installed under the name `network`)."""
import importlib
import logging

import torch

NUM_CLASSES = 19      # datasets.num_classes (/root/reference/datasets/__init__.py:25)


def get_net(args, criterion, criterion_aux=None):
    try:
        import datasets
        num_classes = datasets.num_classes
    except Exception:
        num_classes = NUM_CLASSES
    net = get_model(args=args, num_classes=num_classes, criterion=criterion, criterion_aux=criterion_aux)
    num_params = sum([param.nelement() for param in net.parameters()])
    logging.info('Model params = {:2.3f}M'.format(num_params / 1000000))
    if not torch.cuda.is_available():
        raise RuntimeError('pinthememory_amd has no CPU path: a ROCm GPU (MI355X) and libpinmem_hip.so are required')
    net = net.cuda()
    return net


def warp_network_in_dataparallel(net, gpuid):
    return torch.nn.parallel.DistributedDataParallel(net, device_ids=[gpuid], find_unused_parameters=True)


def get_model(args, num_classes, criterion, criterion_aux=None):
    network = args.arch
    module = network[:network.rfind('.')]
    model = network[network.rfind('.') + 1:]
    mod = importlib.import_module(module)
    net_func = getattr(mod, model)
    return net_func(args=args, num_classes=num_classes, criterion=criterion, criterion_aux=criterion_aux)
