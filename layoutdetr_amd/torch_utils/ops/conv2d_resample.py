"""`conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)` with the reference's signature and
semantics (torch_utils/ops/conv2d_resample.py:47-142): a 2-D convolution combined with FIR up-/down-sampling, padding applied
once with respect to the upsampled image.  Every branch lands on `conv2d_gradfix` (engine) + `upfirdn2d` (FIR kernel); the
StyleGAN2 layers of the hot path use fused forms of the same two branches (hip/modconv.py).
"""
import torch

from . import conv2d_gradfix, upfirdn2d


def _conv(x, w, stride=1, padding=0, transpose=False, correlate=True):
    """torch's conv ops correlate; a true convolution mirrors the taps first."""
    if not correlate and (w.shape[2] > 1 or w.shape[3] > 1):
        w = w.flip([2, 3])
    fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return fn(x, w, stride=stride, padding=padding)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    if groups != 1:
        raise NotImplementedError('conv2d_resample: groups > 1 is not implemented on the gfx950 path')
    kh, kw = int(w.shape[2]), int(w.shape[3])
    fw, fh = upfirdn2d._get_filter_size(f)
    x0, x1, y0, y1 = upfirdn2d._parse_padding(padding)
    # the FIR's own footprint, split around the sample it is centred on
    if up > 1:
        x0 += (fw + up - 1) // 2; x1 += (fw - up) // 2
        y0 += (fh + up - 1) // 2; y1 += (fh - up) // 2
    if down > 1:
        x0 += (fw - down + 1) // 2; x1 += (fw - down) // 2
        y0 += (fh - down + 1) // 2; y1 += (fh - down) // 2
    pointwise = kh == 1 and kw == 1

    if pointwise and up == 1 and down > 1:          # decimate first: the 1x1 conv then runs on 1/down^2 of the pixels
        x = upfirdn2d.upfirdn2d(x, f, down=down, padding=[x0, x1, y0, y1], flip_filter=flip_filter)
        return _conv(x, w, correlate=flip_weight)
    if pointwise and up > 1 and down == 1:          # convolve first: the 1x1 conv runs before the pixel count grows
        x = _conv(x, w, correlate=flip_weight)
        return upfirdn2d.upfirdn2d(x, f, up=up, padding=[x0, x1, y0, y1], gain=up ** 2, flip_filter=flip_filter)
    if up == 1 and down > 1:                        # low-pass at full resolution, then a strided convolution
        x = upfirdn2d.upfirdn2d(x, f, padding=[x0, x1, y0, y1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, correlate=flip_weight)
    if up > 1:                                      # transposed convolution does the zero insertion, the FIR interpolates
        x0 -= kw - 1; x1 -= kw - up; y0 -= kh - 1; y1 -= kh - up
        tx, ty = max(min(-x0, -x1), 0), max(min(-y0, -y1), 0)      # what the transposed conv can crop itself
        x = _conv(x, w.transpose(0, 1), stride=up, padding=[ty, tx], transpose=True, correlate=not flip_weight)
        x = upfirdn2d.upfirdn2d(x, f, padding=[x0 + tx, x1 + tx, y0 + ty, y1 + ty], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if x0 == x1 and y0 == y1 and x0 >= 0 and y0 >= 0:   # plain convolution with symmetric padding
        return _conv(x, w, padding=[y0, x0], correlate=flip_weight)
    # asymmetric / negative padding: let the FIR stage (identity taps) pad or crop, then convolve unpadded
    x = upfirdn2d.upfirdn2d(x, None, padding=[x0, x1, y0, y1], flip_filter=flip_filter)
    return _conv(x, w, correlate=flip_weight)
