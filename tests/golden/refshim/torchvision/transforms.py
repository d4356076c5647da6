"""stand-in: the reference imports these names and never calls them on the degradation path"""
ToPILImage = Compose = RandomCrop = ToTensor = Grayscale = None
