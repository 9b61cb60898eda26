"""Disk-mappable objects: same interface as ``zuds/file.py:16-95``, DB-free."""
from pathlib import Path

__all__ = ['UnmappedFileError', 'File']


class UnmappedFileError(FileNotFoundError):
    """Raised when a method needs the object to be mapped to a file on disk and
    it is not (``zuds/file.py:8-13``)."""
    pass


class File(object):
    """A python object mappable to a file on disk.  The user associates objects
    with files through ``map_to_local_file``; properties describe what is in
    memory, ``save()`` synchronises to disk."""

    __diskmapped_cached_properties__ = ['_path']

    @property
    def local_path(self):
        try:
            return self._path
        except AttributeError:
            raise UnmappedFileError(
                f'File "{getattr(self, "basename", None)}" is not mapped to the local '
                f'file system. Identify the file corresponding to this object on '
                f'the local file system, then call `map_to_local_file`.')

    @property
    def ismapped(self):
        return hasattr(self, '_path')

    def map_to_local_file(self, path, quiet=True):
        if not quiet:
            print(f'Mapping {getattr(self, "basename", self)} to {path}')
        self._path = str(Path(path).absolute())

    def unmap(self):
        if not self.ismapped:
            raise UnmappedFileError(f"Cannot unmap file '{getattr(self, 'basename', None)}', "
                                    f"file is not mapped")
        self.clear()

    def clear(self):
        for attr in self.__diskmapped_cached_properties__:
            if hasattr(self, attr):
                delattr(self, attr)

    def save(self):
        raise NotImplementedError

    def load(self):
        raise NotImplementedError
