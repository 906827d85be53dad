"""Mirror of /root/reference/pytorch/TrainableJacobiDiag: the learned Jacobi diagonal (model inference, the
Gelfand-formula damping-factor loss and its backward).  Datasets, plotting and the training script's epoch loop are
the caller's; `loss_batch(model, batch)` + `loss.backward()` + a torch optimiser is a complete training step."""
from . import TrainableJacobiGNN, loss  # noqa: F401
from .TrainableJacobiGNN import get_model  # noqa: F401
