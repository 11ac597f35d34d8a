"""The config-2 workload generator: glcGenZipfPhilox on the device == tests/datagen.zipf_philox_bytes on the
host, for any byte range -- so every block of the 4 GiB bench input can be regenerated on the CPU (bench.py does, for the
blocks it compares with the oracle)."""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


def test_device_zipf_stream_equals_host_stream(glc, cuda):
    import torch
    L = glc.lib()
    thr = torch.from_numpy(datagen.zipf_thresholds().view(np.int32)).to(cuda)
    for first, n in ((0, 1 << 20), (3 << 20, 1 << 18), ((4095 << 20) + 4096, 65536), ((1 << 33) + 16, 4096)):
        out = torch.zeros(n, dtype=torch.uint8, device=cuda)
        assert L.glcGenZipfPhilox(out.data_ptr(), n, first, 0x5EED0002, thr.data_ptr(), None) == 1
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), datagen.zipf_philox_bytes(first, n)), (first, n)
    out = torch.zeros(32, dtype=torch.uint8, device=cuda)
    assert L.glcGenZipfPhilox(out.data_ptr(), 24, 0, 1, thr.data_ptr(), None) == 0          # not a multiple of 16
    assert L.glcGenZipfPhilox(None, 16, 0, 1, thr.data_ptr(), None) == 0


def test_device_float_stream_equals_host_stream(glc, cuda):
    """config 4: the same float32 BITS on both sides (an integer sum, exact steps and one correctly rounded multiply)"""
    import torch
    L = glc.lib()
    for first, n in ((0, 1 << 20), (7 << 20, 1 << 16), ((1 << 34) + 32, 4096)):
        out = torch.zeros(n, dtype=torch.uint8, device=cuda)
        assert L.glcGenFloatPhilox(out.data_ptr(), n, first, 0x5EED0004, None) == 1
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), datagen.float_philox_bytes(first, n)), (first, n)
    f = datagen.float_philox_bytes(0, 1 << 20).view(np.float32)
    assert abs(float(f.mean())) < 0.01 and abs(float(f.std()) - 1.0) < 0.01

