import torch, ctypes
x = torch.arange(16, dtype=torch.float32, device='cuda:0')
class Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2, 'strides': None}
v = torch.as_tensor(Raw(x.data_ptr() + 16, 8), device='cuda:0')
print('view', v.tolist(), v.data_ptr() == x.data_ptr() + 16)
v += 100
print('orig', x.tolist())
import torch.distributed as dist
print('nccl available', dist.is_nccl_available(), 'devices', torch.cuda.device_count())
