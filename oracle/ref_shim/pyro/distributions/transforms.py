from torch.distributions.transforms import *  # noqa: F401,F403
from torch.distributions.transforms import Transform  # noqa: F401
from torch.distributions import biject_to  # noqa: F401
