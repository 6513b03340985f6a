"""GPU: error behaviour of the C ABI (every misuse is a PINN_E* code with a message, never a crash): the reference
would raise from Keras / numpy shape checks at the same call sites."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LB, UB = np.array([-1.0, 0.0]), np.array([1.0, 0.99])


def test_misuse_raises_with_a_message():
    import pinn_native
    E = pinn_native.PinnNativeError
    eng = pinn_native.Engine([2, 20, 20, 1], LB, UB, pde="burgers", dtype="f32")
    with pytest.raises(E, match="not a discrete-time model"):
        eng.disc_set_stage(0, np.zeros(3), np.zeros(3), None)
    with pytest.raises(E, match="not a discrete-time model"):
        eng.disc_predict(0, np.zeros(3))
    with pytest.raises(E, match="Schrodinger-only"):
        eng.set_boundary(np.zeros((2, 2)), np.zeros((2, 2)))
        eng.set_collocation(np.zeros((4, 2)))
        eng.loss_grad()
    eng.close()
    eng = pinn_native.Engine([2, 20, 20, 1], LB, UB, pde="burgers_ide", dtype="f32")
    with pytest.raises(E, match="no collocation set"):
        eng.lhs_collocation(100, seed=1)
    with pytest.raises(E, match="mailboxes are not attached"):
        eng.comm_set_mode("mailbox")
    with pytest.raises(E):
        eng.set_weights(np.zeros(5))
    eng.close()
    with pytest.raises(E, match="hidden widths must be equal"):
        pinn_native.Engine([2, 20, 30, 1], LB, UB, pde="burgers")
    with pytest.raises(E, match="one input"):
        pinn_native.Engine([2, 20, 20, 5], LB, UB, pde="burgers_disc")
    with pytest.raises(E, match="output size"):
        pinn_native.Engine([2, 20, 20, 2], LB, UB, pde="burgers")
    d = pinn_native.Engine([1, 20, 20, 5], [-1.0], [1.0], pde="burgers_disc", dtype="f64")
    with pytest.raises(E, match="stage sets"):
        d.set_collocation(np.zeros((4, 2)))
    with pytest.raises(E, match="no stage set"):
        d.loss_grad()
    with pytest.raises(E, match="outside 1..n_out"):
        d.disc_set_stage(0, np.zeros(3), np.zeros(3), np.zeros((5, 9)))
    with pytest.raises(E, match="different q"):
        d.disc_set_stage(0, np.zeros(3), np.zeros(3), np.zeros((5, 4)))
        d.disc_set_stage(1, np.zeros(3), np.zeros(3), np.zeros((5, 3)))
        d.loss_grad()
    d.close()
