"""The pieces together (window assembly -> forward -> fused MSE backward -> Adam -> metrics), on device: the loss of a
learnable synthetic task goes down."""
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_flat_training_loop_learns(dtype):
    from examples.train_flat import train
    losses, rate = train(steps=60, batch=512, dtype=dtype, layers=2, lr=2e-3, rows=20_000, log_every=20, quiet=True)
    first, last = losses[0][1], losses[-1][1]
    assert last < 0.7 * first, losses
    assert abs(losses[-1][2] ** 2 - last) <= 1e-3 * last      # StepMetrics' RMSE^2 == the fused kernel's MSE of the same step


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_wrapper_training_loop_learns(dtype):
    """The same task through the reference's wrapper API (training_step -> zero_grad -> backward -> torch.optim.Adam on the wrapper's
    parameters, Lightning's order): the loss goes down and the logged RMSE is the square root of the logged MSE."""
    import os
    from examples.train_wrapper import train
    prev = os.environ.get("MSHGNN_DTYPE")
    try:
        losses, rate = train(steps=60, batch=512, dtype=dtype, layers=2, lr=2e-3, rows=20_000, log_every=20, quiet=True)
    finally:
        if prev is None:
            os.environ.pop("MSHGNN_DTYPE", None)
        else:
            os.environ["MSHGNN_DTYPE"] = prev
    first, last = losses[0][1], losses[-1][1]
    assert last < 0.7 * first, losses
    assert abs(losses[-1][2] ** 2 - last) <= 1e-3 * last
