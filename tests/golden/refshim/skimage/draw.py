disk = None
