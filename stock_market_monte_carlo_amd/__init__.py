"""MI355X-native Monte-Carlo returns engine: a drop-in for the Monte-Carlo path of
matthijsvk/stock_market_monte_carlo (src/simulations.{cpp,cu}).  See DESIGN.md."""
from ._lib import MODE_GAUSSIAN, MODE_TABLE, SmmcError  # noqa: F401
from .engine import (Engine, Group, Stats, many_updates, mc_simulations, mc_simulations_gpu,  # noqa: F401
                     mc_simulations_gpu_reduceBlock, mc_simulations_keepdata, read_historical_returns,
                     reduce_mean_gpu, update_count_below_min, update_fund, update_mean_std, update_quartiles,
                     vector_add_gpu)
