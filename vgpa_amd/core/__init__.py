"""Drop-in counterparts of the reference's ``core`` package."""
