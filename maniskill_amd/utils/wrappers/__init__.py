from .record import RecordEpisode
