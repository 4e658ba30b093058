"""Drop-in counterparts of the reference's ``dynamics`` package (``energy_mode``)."""
