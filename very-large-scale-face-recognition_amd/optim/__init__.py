from .optimizer import get_optim_scheduler
