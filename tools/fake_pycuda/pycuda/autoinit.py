from .driver import Context
context = Context.get_current()
