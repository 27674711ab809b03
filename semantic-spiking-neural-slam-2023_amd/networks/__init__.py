"""Network builders with the reference's names (``sspslam/networks/__init__.py:2-9``)."""
from .pathintegration import PathIntegration, get_to_Fourier, get_from_Fourier, make_feedback  # noqa: F401
from .binding import CircularConvolution, Product, circconv, transform_in, transform_out, dft_half  # noqa: F401
from .associativememory import AssociativeMemory  # noqa: F401
from .slam import SLAMNetwork, get_slam_input_functions, get_slam_input_functions2  # noqa: F401
from .slam_view import SLAMViewNetwork, get_slamview_input_functions  # noqa: F401
