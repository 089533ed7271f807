"""The drop-in boundary's conventions (reference style_transfer.py:310-347,373-374,405,467,487-493 and
cli.py:124-133,261-266): attributes callers read, Python exceptions for bad arguments, the callback may call
get_image*() re-entrantly, and a KeyboardInterrupt between two iterations leaves get_image() working."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pil(seed, h, w):
    from PIL import Image
    g = torch.Generator().manual_seed(seed)
    return Image.fromarray((torch.rand((h, w, 3), generator=g).numpy() * 255).astype(np.uint8), 'RGB')


@pytest.fixture(scope='module')
def st(vgg_weights):
    import style_transfer as st_pkg
    return st_pkg.StyleTransfer(devices=[DEV], weights=vgg_weights)


def test_attributes_of_a_fresh_object(st):
    assert [str(d) for d in st.devices] == [DEV]
    assert st.content_layers == [22] and st.style_layers == [1, 6, 11, 20, 29]
    assert len(st.style_weights) == 5 and abs(sum(abs(w) for w in st.style_weights) - 1.0) < 1e-6
    assert st.image is None and st.average is None and st.model is not None
    assert st.get_image() is None                              # before the first scale (reference :338-339)


def test_argument_errors_are_python_exceptions(st, vgg_weights):
    import style_transfer as st_pkg
    content, style = _pil(1, 48, 40), _pil(2, 40, 44)
    with pytest.raises(ValueError):                            # mismatched style weights (:373-374)
        st.stylize(content, [style, style], style_weights=[1.0], min_scale=32, end_scale=32, initial_iterations=1)
    with pytest.raises(ValueError):                            # unknown init (:405)
        st.stylize(content, [style], init='noise', min_scale=32, end_scale=32, initial_iterations=1)
    with pytest.raises(ValueError):                            # unknown optimiser (:467)
        st.stylize(content, [style], optimizer='sgd', min_scale=32, end_scale=32, initial_iterations=1)
    with pytest.raises(ValueError):                            # input smaller than the network allows (:81-83)
        st.stylize(_pil(3, 12, 12), [style], min_scale=8, end_scale=8, initial_iterations=1)
    with pytest.raises(ValueError):                            # 'Only 1 or 2 devices are supported.' (:331) - here 1 to 8
        st_pkg.StyleTransfer(devices=[DEV] * 9, weights=vgg_weights)
    st_pkg.StyleTransfer(devices=[DEV, DEV, DEV], weights=vgg_weights)      # (round 5: a device list is the in-process sharding)
    with pytest.raises(KeyError):                              # unknown pooling: a KeyError like the reference's table lookup (:38)
        st_pkg.StyleTransfer(devices=[DEV], pooling='median', weights=vgg_weights)


def test_callback_may_read_the_image_and_an_interrupt_keeps_it(st):
    content, style = _pil(4, 64, 56), _pil(5, 48, 72)
    seen = []

    def callback(it):
        # cli.py:124-133: the progress callback saves / displays the current image while the optimisation runs
        t = st.get_image_tensor()
        assert t.shape == (3, it.h, it.w) and float(t.min()) >= 0.0 and float(t.max()) <= 1.0
        assert isinstance(it.loss, float) and isinstance(it.gpu_ram, int) and it.time > 0
        seen.append((it.w, it.h, it.i, it.i_max))
        if len(seen) == 7:
            raise KeyboardInterrupt                           # cli.py:261-266: Ctrl-C between two iterations

    with pytest.raises(KeyboardInterrupt):
        st.stylize(content, [style], min_scale=32, end_scale=64, initial_iterations=5, iterations=5, callback=callback)
    assert len(seen) == 7 and seen[0][2:] == (1, 5) and seen[5][2] == 1 and seen[5][:2] != seen[4][:2]
    im = st.get_image()                                        # still works, at the interrupted scale's size
    assert im.size == (seen[-1][0], seen[-1][1])
    arr = st.get_image('np_uint16')
    assert arr.dtype == np.uint16 and arr.shape == (seen[-1][1], seen[-1][0], 3)
    with pytest.raises(ValueError):
        st.get_image('jpeg')
    # and the object can be used again afterwards
    out = st.stylize(content, [style], min_scale=32, end_scale=32, initial_iterations=2)
    assert out.size == st.get_image().size


@pytest.mark.parametrize('size', [256, 1024])
def test_stream_layouts_and_repeated_closures_give_identical_results(size, vgg_weights):
    """The closure's side streams are chosen by a hardware-queue probe (round 4, shared_head_streams); which stream carries
    which head - and whether foreign streams existed first - must not change a single bit, and neither may repeating the
    closure (a kernel that races with another stream's work shows up as run-to-run differences: the TV term did, when a
    first version of the compact layout ran it beside the backward trunk)."""
    from style_transfer import _hip as hip
    import st_oracle as O
    g = torch.Generator().manual_seed(size)
    low = torch.rand((3, 1, 3, size // 16, size // 16), generator=g)
    content, style, image = (torch.nn.functional.interpolate(t, (size, size), mode='bicubic').clamp(0, 1) for t in low)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    results = {}
    for compact in (1, 0):
        for lockstep in (1, 0):
            with hip.options(ST_STREAMS_COMPACT=compact, ST_HEAD_LOCKSTEP=lockstep, ST_STREAM_DUMMIES=2 if compact else 0):
                plan = hip.Plan(net, size, size)
                plan.forward(content.to(DEV), 22)
                plan.set_content_target_from_forward()
                plan.forward(style.to(DEV), 29)
                for i, layer in enumerate(O.STYLE_LAYERS):
                    plan.set_style_target(i, *plan.moments(layer))
                plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
                seen = set()
                for _ in range(6):
                    losses, grad = plan.loss_and_grad(image.to(DEV))
                    torch.cuda.synchronize()
                    seen.add((tuple(losses.cpu().tolist()), float(grad.double().abs().sum())))
                assert len(seen) == 1, f'compact={compact} lockstep={lockstep}: {len(seen)} different results in 6 closures'
                results[(compact, lockstep)] = seen.pop()
                del plan
    for lockstep in (1, 0):
        assert results[(1, lockstep)] == results[(0, lockstep)], 'the stream layout changed the result'
