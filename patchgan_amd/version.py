__version__ = '0.2.2+mi355x.1'
