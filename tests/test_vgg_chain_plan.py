"""Host logic of the bf16-mode loss-VGG chain (autograd_vgg.py): the unit plan and the capture rule of
IntermediateLayerGetter over a network with in-place ReLUs (trainer.py:165-199).  No GPU."""
from style_transfer_amd.autograd_vgg import VGGChainFunction, capture_sets, plan_units
from style_transfer_amd.loss_network import vgg19_layers


def test_units_pair_a_conv_with_its_relu():
    layout = vgg19_layers()
    units = plan_units(layout, 25)
    assert units[0] == ("conv", 0, True) and units[1] == ("conv", 2, True) and units[2] == ("pool", 4)
    assert units[-1] == ("conv", 25, False)                 # the network ends on a conv: no ReLU follows
    assert [u[1] for u in units if u[0] == "pool"] == [4, 9, 18]
    assert plan_units(layout, 26)[-1] == ("conv", 25, True)  # ... one index later its ReLU is part of the unit
    assert plan_units(layout, 4)[-1] == ("pool", 4)


def test_capture_rule_cli_defaults():
    layout = vgg19_layers()
    pre, post = capture_sets(layout, [[21], [2, 7, 14, 19, 25]])
    assert pre == {21, 25}                                   # the last index of each extractor, before its ReLU
    assert post == {2, 7, 14, 19}                            # overwritten by the in-place ReLU that follows
    keys = VGGChainFunction.keys(layout, 25, pre, post)
    assert keys == [("post", 2), ("post", 7), ("post", 14), ("post", 19), ("pre", 21), ("pre", 25)]


def test_capture_rule_relu_and_pool_indices_and_double_capture():
    layout = vgg19_layers()
    pre, post = capture_sets(layout, [[4, 8], [1, 6, 11]])
    assert pre == set() and post == {4, 8, 1, 6, 11}         # 8 and 11 are ReLU indices: nothing is captured pre-ReLU
    keys = VGGChainFunction.keys(layout, 11, pre, post)
    # ReLU indices 1, 6, 8, 11 are served by the conv units 0, 5, 7, 10; 4 is the pool's own output
    assert keys == [("post", 0), ("post", 4), ("post", 5), ("post", 7), ("post", 10)]
    pre, post = capture_sets(layout, [[25], [21, 25]])
    assert pre == {25} and post == {21}
    pre, post = capture_sets(layout, [[21], [21, 25]])
    assert pre == {21, 25} and post == {21}                  # conv 21 both before (set 1) and after (set 2) its ReLU
    assert VGGChainFunction.keys(layout, 25, pre, post) == [("pre", 21), ("post", 21), ("pre", 25)]
