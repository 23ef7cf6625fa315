
from fourm import _upstream as _up

_up.extend_path(__name__, __path__)
