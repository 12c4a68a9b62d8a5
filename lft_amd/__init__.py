"""lft_amd: MI355X-native forward of the LFT light-field super-resolution network."""
from .module import get_loss, get_model, weights_init  # noqa: F401
