
from . import meta_arch  # noqa: F401,E402
