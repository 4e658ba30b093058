"""Drop-in counterparts of the reference's ``auxiliary`` helpers used on the hot path."""
